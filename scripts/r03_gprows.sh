#!/bin/bash
# key ranges per frame of the one-frame gated attentions (local window, self) with 8 clips per launch: workload A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
for e in "X=0" "RMEM_GP_LOCAL_ROWS=4" "RMEM_GP_LOCAL_ROWS=2" "RMEM_GP_SELF_ROWS=4" "RMEM_GP_SELF_ROWS=2" "X=0"; do
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$e $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
