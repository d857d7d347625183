#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 1024 1536 2048; do echo "WGS=$w: $(RMEM_STEM_POOL_WGS=$w timeout -k 10 200 python scripts/stem_bench.py 2>&1 | grep 'one pass')"; done
for a in "X=0" "RMEM_STEM_POOL_WGS=1024" "RMEM_STEM_POOL_WGS=1536" "X=0" "RMEM_STEM_POOL_WGS=1024"; do
  echo "== $a: $(env $a timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>&1 | tail -1)"
done
