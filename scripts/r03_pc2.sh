#!/bin/bash
# producer / consumer form for the dual-source (conv3 + shortcut) GEMM: tests under the switch, then pipeline A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03pc
mkdir -p $O
RMEM_GEMM_PC_DUAL=2 timeout -k 10 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "dual or conv2d or bneck" > $O/t.txt 2>&1 || { tail -20 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for e in "X=0" "RMEM_GEMM_PC_DUAL=2" "RMEM_GEMM_PC_DUAL=3" "X=0" "RMEM_GEMM_PC_DUAL=2" "RMEM_GEMM_PC_DUAL=3"; do
  env $e timeout -k 10 240 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$e $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
