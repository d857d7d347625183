#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
for v in "RMEM_STREAM_ORDER=pairs" "RMEM_STREAM_ORDER=mains_first" "RMEM_STREAM_ORDER=encs_first" "RMEM_STREAM_ORDER=pairs" "RMEM_STREAM_ORDER=mains_first" "RMEM_STREAM_ORDER=mains_first GPU_MAX_HW_QUEUES=6" "RMEM_STREAM_ORDER=mains_first GPU_MAX_HW_QUEUES=8" "RMEM_STREAM_ORDER=pairs GPU_MAX_HW_QUEUES=2"; do
  echo "== $v"
  env $v timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', j['value'])" || { tail -20 $O/err.txt; exit 1; }
done
