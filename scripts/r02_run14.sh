#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attn or attention or mem_read" > gpurun_out/r2_t14.log 2>&1
rc=$?
tail -4 gpurun_out/r2_t14.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 120 python scripts/attn_bench.py --T 8 --iters 20 --wgs 448,896,1792,3584 2>&1 | grep "T=8" | tee gpurun_out/r2_pipe.txt
timeout -k 10 120 python scripts/attn_bench.py --T 1,2,4,6 --iters 20 2>&1 | grep "T=" | tee -a gpurun_out/r2_pipe.txt
