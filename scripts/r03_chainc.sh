#!/bin/bash
# chain C with the second staging buffer inside dead A panels (97 KB instead of 130 KB of LDS): identity tests, then pipeline A/B is by commit
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/q
timeout -k 10 500 python -m pytest tests/test_hip_ops.py tests/test_hip_engine.py -x -q -m gpu -k "chain or group_engine_matches or lstt" > gpurun_out/q/t.txt 2>&1 || { tail -30 gpurun_out/q/t.txt; exit 1; }
tail -2 gpurun_out/q/t.txt
for e in "X=0" "RMEM_LIB_PATH=$GRAFT_REPO_ROOT/rmem_ocu_amd/librmem_prev.so" "X=0" "RMEM_LIB_PATH=$GRAFT_REPO_ROOT/rmem_ocu_amd/librmem_prev.so"; do
  env $e timeout -k 10 240 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/q/c.txt 2>&1 || { tail -5 gpurun_out/q/c.txt; exit 1; }
  echo "${e:0:14} $(python -c "import json,sys; d=json.loads(open('gpurun_out/q/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
