#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
for v in "4 24" "6 24" "8 24" "8 32" "12 24" "6 36" "4 24"; do
  set -- $v
  echo "== clips per group $1, clips in flight $2"
  timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 --clips-per-group $1 --clips-in-flight $2 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', j['value'], 'frames/step', j['config']['frames_per_step'])" || { tail -20 $O/err.txt; exit 1; }
  timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 --clips-per-group $1 --clips-in-flight $2 --steps 20 --warmup 5 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form', j['value'])" || { tail -20 $O/err.txt; exit 1; }
done
