#!/bin/bash
# DeAOT GPU tests after a host-side change
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
timeout -k 10 500 python -m pytest tests/test_hip_deaot_ops.py tests/test_hip_deaot_engine.py -x -q -m gpu > gpurun_out/q/t.txt 2>&1 || { tail -30 gpurun_out/q/t.txt; exit 1; }
tail -2 gpurun_out/q/t.txt
