#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_engine.py -m gpu -q -x -s -k "cfg5_swin" > $O/t.log 2>&1
rc=$?
grep -v "^$" $O/t.log | tail -12 | cut -c1-300
exit $rc
