#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests_head.log 2>&1
rc=$?
tail -3 $O/gpu_tests_head.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 | cut -c1-120
