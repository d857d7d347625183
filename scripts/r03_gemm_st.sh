#!/bin/bash
# per-layer GEMM timing with deeper LDS rings for the 128x128 tile (few-tile, deep-K problems: ResNet layer 3, decoder 16x)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g
mkdir -p $O
for st in 3 4 5; do
  RMEM_GEMM_BIG_DEEP_ST=$st timeout -k 10 200 python scripts/gemm_bench.py --iters 30 > $O/gemm_st$st.txt 2>&1 || { tail -5 $O/gemm_st$st.txt; exit 1; }
done
paste -d'|' $O/gemm_st3.txt $O/gemm_st4.txt $O/gemm_st5.txt | cut -c1-230
