#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "X=0" "RMEM_CU_SPLIT=1/4" "RMEM_CU_SPLIT=1/4/all" "RMEM_CU_SPLIT=2/4" "RMEM_CU_SPLIT=2/4/all" "RMEM_CU_SPLIT=3/8" "RMEM_CU_SPLIT=4/4/all" "X=0"; do
  echo "== $a: $(env $a timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>&1 | tail -1)"
done
