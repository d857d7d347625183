#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r2_sweep3.txt
for pc in 1 2 4 1; do
  v=$(RMEM_PLAIN_CHUNKS=$pc timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "RMEM_PLAIN_CHUNKS=$pc: $v frames/s" >> gpurun_out/r2_sweep3.txt
done
for w in 896 1792 2688 3584; do
  v=$(RMEM_ATTN_WGS=$w timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "RMEM_ATTN_WGS=$w: $v frames/s" >> gpurun_out/r2_sweep3.txt
done
cat gpurun_out/r2_sweep3.txt
timeout -k 10 600 python -m pytest tests/test_hip_engine.py tests/test_hip_ops.py -m gpu -q -k "race or iou_counts" 2>&1 | tail -3
