#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "X=0" "RMEM_ATTN_WGS=896" "RMEM_ATTN_WGS=3584" "X=0" "RMEM_ATTN_WGS=896"; do
  echo "== $a: $(env $a timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 8 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])")"
done
