#!/bin/bash
# clips per group x groups on the final tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q
mkdir -p $O
run() { timeout -k 10 300 python bench.py --steps 60 --warmup 12 --no-cpu-baseline "$@" > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$* $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['frames_per_step'])")"; }
run &&
run --clips-per-group 12 --clips-in-flight 36 &&
run --clips-per-group 12 --clips-in-flight 24 &&
run --clips-per-group 10 --clips-in-flight 30 &&
run --clips-per-group 16 --clips-in-flight 32 &&
run
