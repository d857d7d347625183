#!/bin/bash
# per-layer GEMM timing at the round-3 batch sizes (16 images per encoder launch, 8 clips per LSTT / decoder launch) under the tile /
# ring-depth switches of gemm_conv.hip; one process per setting (the switches are read once)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u
mkdir -p $O
i=0
for env in "X=0" "RMEM_GEMM_BIG_DEEP=512" "RMEM_GEMM_BIG_DEEP=4096" "RMEM_GEMM_BIG_ST=2" "RMEM_GEMM_BIG=0" "RMEM_GEMM_ST=2" "RMEM_GEMM_BIG_K=128" "RMEM_GEMM_BIG64=1"; do
  env $env timeout -k 10 200 python scripts/gemm_bench.py --no-swin > $O/v$i.txt 2>&1 || { tail -5 $O/v$i.txt; exit 1; }
  echo "v$i = $env: $(tail -1 $O/v$i.txt)"
  i=$((i+1))
done
python - <<'PY'
import glob
cols=[]
for i in range(8):
    L=[l for l in open(f'gpurun_out/r03u/v{i}.txt') if l.startswith('conv')]
    cols.append(L)
for r in range(len(cols[0])):
    name=cols[0][r].split(':')[0]
    print(f'{name:40s}', ' '.join(f"{float(c[r].split(':')[1].split('us')[0]):7.1f}" for c in cols))
PY
