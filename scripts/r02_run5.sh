#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r2_ablate.txt
for v in "" OCC2 OCC1; do
  echo "=== variant '$v'" >> gpurun_out/r2_ablate.txt
  if [ -n "$v" ]; then export RMEM_LIB_PATH=$GRAFT_REPO_ROOT/build_variants/librmem_$v.so; else unset RMEM_LIB_PATH; fi
  timeout -k 10 120 python scripts/attn_bench.py --T 8 --iters 20 --wgs 1792,3584 2>&1 | grep "T=8" >> gpurun_out/r2_ablate.txt
done
cat gpurun_out/r2_ablate.txt
