#!/bin/bash
# occupancy experiment for the memory-read kernel: dynamic LDS padding limits the workgroups per CU (33.5 KB static per workgroup)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r2_occ.txt
for pad in 0 8000 21000 48000 100000; do
  echo "=== RMEM_ATTN_LDS_PAD=$pad" >> gpurun_out/r2_occ.txt
  RMEM_ATTN_LDS_PAD=$pad timeout -k 10 120 python scripts/attn_bench.py --T 8 --iters 20 --wgs 1792,3584 2>&1 | grep "T=8" >> gpurun_out/r2_occ.txt
done
cat gpurun_out/r2_occ.txt
