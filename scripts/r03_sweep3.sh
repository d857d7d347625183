#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
for v in "8 24 2" "12 36 2" "16 48 2" "16 32 2" "10 40 2" "12 48 2" "8 24 2"; do
  set -- $v
  echo "== clips per group $1, clips in flight $2, lookahead $3"
  timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 --clips-per-group $1 --clips-in-flight $2 --encoder-lookahead $3 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', j['value'], 'frames/step', j['config']['frames_per_step'])" || { tail -20 $O/err.txt; exit 1; }
done
