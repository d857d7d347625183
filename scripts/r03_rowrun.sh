#!/bin/bash
# row-run GEMM forms (stem 7x7x8, id bank 17x17x16) on 128-row tiles: parity of the conv tests under the switch, then timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v
mkdir -p $O
RMEM_GEMM_ROWRUN_BIG=3 timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "test_conv2d" > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for env in "X=0" "RMEM_GEMM_ROWRUN_BIG=3" "RMEM_GEMM_ROWRUN_BIG=3 RMEM_GEMM_ROWRUN_ST=2" "RMEM_GEMM_ROWRUN_BIG=3 RMEM_GEMM_ROWRUN_ST=1" "RMEM_GEMM_ROWRUN_BIG=1 RMEM_GEMM_ROWRUN_ST=4"; do
  echo "== $env"
  env $env timeout -k 10 200 python scripts/gemm_bench.py --only-rowrun 2>&1 | grep "^conv"
done
for env in "X=0" "RMEM_GEMM_ROWRUN_BIG=3" "RMEM_GEMM_ROWRUN_BIG=1" "RMEM_GEMM_ROWRUN_BIG=2"; do
  echo "== $env"
  env $env timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 4 | cut -c1-140
done
