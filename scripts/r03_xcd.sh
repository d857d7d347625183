#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "conv or linear or dual or grouped or bank" > $O/tests.log 2>&1
rc=$?
tail -5 $O/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for x in 0 1; do
  RMEM_GEMM_XCD=$x timeout -k 10 200 python scripts/gemm_bench.py --iters 30 > $O/gemm_xcd$x.txt 2>&1 || { tail -5 $O/gemm_xcd$x.txt; exit 1; }
done
paste -d'|' $O/gemm_xcd0.txt $O/gemm_xcd1.txt | cut -c1-160
for x in 0 1 0 1; do
  echo "== RMEM_GEMM_XCD=$x"
  RMEM_GEMM_XCD=$x timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', j['value'])" || { tail -20 $O/err.txt; exit 1; }
done
