#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
for pc in 0 2 3; do
  RMEM_GEMM_PC=$pc timeout -k 10 200 python scripts/gemm_bench.py > gpurun_out/q/g$pc.txt 2>&1 || { tail -5 gpurun_out/q/g$pc.txt; exit 1; }
done
python - <<'PY'
cols=[[l for l in open(f'gpurun_out/q/g{i}.txt') if l.startswith('conv')] for i in (0,2,3)]
for r in range(len(cols[0])):
    name=cols[0][r].split(':')[0]
    if '14400' in name or '3600' in name:
        print(f'{name:40s}', ' '.join(f"{float(c[r].split(':')[1].split('us')[0]):7.1f}" for c in cols), cols[1][r].split('us')[1].strip())
PY
