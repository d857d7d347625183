#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "producer_consumer or conv2d" > gpurun_out/q/t.txt 2>&1 || { tail -30 gpurun_out/q/t.txt; exit 1; }
tail -3 gpurun_out/q/t.txt
