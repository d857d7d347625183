#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r2_t8.log 2>&1
rc=$?
tail -30 gpurun_out/r2_t8.log
exit $rc
