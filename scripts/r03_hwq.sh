#!/bin/bash
# hardware queues on the final tree (the earlier sweeps predate the in-place kernels and the loader-wave GEMM)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q
mkdir -p $O
for e in "X=0" "GPU_MAX_HW_QUEUES=3" "GPU_MAX_HW_QUEUES=5" "GPU_MAX_HW_QUEUES=2" "X=0"; do
  env $e timeout -k 10 240 python bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$e $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
done
