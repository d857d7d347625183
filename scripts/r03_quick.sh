#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "X=0" "RMEM_GEMM_DEEP_WGS=512" "RMEM_GEMM_DEEP_WGS=2048" "RMEM_GEMM_DEEP_WGS=0" "RMEM_GEMM_BIG=256" "RMEM_GEMM_BIG=64" "RMEM_GEMM_BIG_DEEP=0" "RMEM_GEMM_BIG_DEEP=128" "RMEM_GEMM_ROWRUN_BIG=0" "X=0"; do
  echo "== $a: $(env $a timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>&1 | tail -1)"
done
