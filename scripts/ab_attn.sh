#!/bin/bash
# A/B of memory-read kernel variants on ONE box: the in-tree library against variant builds under experiments/ab/*.so
# (RMEM_LIB_PATH), alternating, 3 rounds each; prints the median call time (attention + merge + mass) of every run.
cd "$(dirname "$0")/.."
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in rmem_ocu_amd/librmem_hip.so experiments/ab/*.so; do
    echo -n "$(basename $lib) : "
    RMEM_LIB_PATH=$PWD/$lib timeout -k 10 120 python scripts/attn_bench.py --T 8 --clips 4 --mass --iters 60 2>/dev/null | tail -1 || exit 1
  done
done
