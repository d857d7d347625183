#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "X=0" "RMEM_NO_CHAIN=1" "X=0" "RMEM_NO_CHAIN=1"; do
  echo "== swin $a: $(env $a timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 --workload lvos_720p_swinb_N12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>&1 | tail -1)"
done
for a in "X=0" "RMEM_NO_CHAIN=1"; do
  echo "== cfg3 $a: $(env $a timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 --workload ytvos_720p_r50_N8_inject 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>&1 | tail -1)"
done
