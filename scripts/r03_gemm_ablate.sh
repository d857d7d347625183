#!/bin/bash
# where does a GEMM launch spend its time?  timing-only ablations (results wrong by construction): 1 no MFMA, 2 no k-loop DMA, 4 no epilogue, 8 no global traffic in the epilogue
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g
mkdir -p $O
for d in 0 1 2 3 4 8 7 0; do
  RMEM_GEMM_DEBUG=$d timeout -k 10 200 python scripts/gemm_bench.py --iters 20 2>/dev/null | grep -v amdgpu.ids | awk -v d=$d '{printf "%s %s %s %s %s %s  dbg%s %6.1f\n", $2,$3,$4,$5,$6,$7,d,$8}' > $O/ab_$d.txt
done
paste $O/ab_0.txt <(awk '{print $NF}' $O/ab_1.txt) <(awk '{print $NF}' $O/ab_2.txt) <(awk '{print $NF}' $O/ab_3.txt) <(awk '{print $NF}' $O/ab_4.txt) <(awk '{print $NF}' $O/ab_8.txt) <(awk '{print $NF}' $O/ab_7.txt) | head -24
echo "columns: full | no MFMA | no DMA | neither | no epilogue | epilogue without global traffic | nothing"
