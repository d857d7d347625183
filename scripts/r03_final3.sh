#!/bin/bash
# last tree of the round: GPU tier + smoke
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final3
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/final3/gpu_tests.log 2>&1
rc=$?
tail -3 gpurun_out/final3/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
