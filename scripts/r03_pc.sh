#!/bin/bash
# producer / consumer (loader-wave) form of the 128x128 LDS-DMA GEMM: bit identity against the default, per-layer timing, pipeline A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pc
mkdir -p $O
for pc in 0 1 2; do
  RMEM_GEMM_PC=$pc timeout -k 10 120 python scripts/pc_check.py > $O/check$pc.txt 2>&1 || { tail -5 $O/check$pc.txt; exit 1; }
done
for pc in 1 2; do cmp $O/check0.txt $O/check$pc.txt && echo "pc=$pc identical"; done
cat $O/check0.txt
for pc in 0 1 2; do
  RMEM_GEMM_PC=$pc timeout -k 10 200 python scripts/gemm_bench.py --no-swin > $O/g$pc.txt 2>&1 || { tail -5 $O/g$pc.txt; exit 1; }
done
python - <<'PY'
cols=[[l for l in open(f'gpurun_out/r03pc/g{i}.txt') if l.startswith('conv')] for i in (0,1,2)]
for r in range(len(cols[0])):
    name=cols[0][r].split(':')[0]
    print(f'{name:40s}', ' '.join(f"{float(c[r].split(':')[1].split('us')[0]):7.1f}" for c in cols))
PY
for pc in 0 1 2 0 1 2 0 1 2; do
  RMEM_GEMM_PC=$pc timeout -k 10 240 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/b$pc.txt 2>&1 || { tail -5 $O/b$pc.txt; exit 1; }
  echo "pc=$pc $(python -c "import json,sys; d=json.loads(open('$O/b$pc.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
