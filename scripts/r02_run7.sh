#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -k "fp16" > gpurun_out/r2_t7a.log 2>&1
tail -15 gpurun_out/r2_t7a.log
timeout -k 10 600 python -m pytest tests/test_hip_engine.py tests/test_stage_budget.py -m gpu -q -s -k "fp16 or stage" > gpurun_out/r2_t7b.log 2>&1
grep -v "^\.*$" gpurun_out/r2_t7b.log | tail -70
