#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_engine.py -m gpu -q -x -k "evaluator or objects or tta or small_clip_teacher" > gpurun_out/r2_t11.log 2>&1
rc=$?
tail -12 gpurun_out/r2_t11.log
exit $rc
