#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 800 python -m pytest tests/test_hip_engine.py -m gpu -q -x -s -k "cfg3_geometry or test_conv2d or new_object" > $O/t.log 2>&1
rc=$?
grep -v "^$" $O/t.log | tail -15 | cut -c1-400
exit $rc
