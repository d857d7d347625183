#!/bin/bash
# gated attention kernels: DeAOT op + engine tests, then the workload A/B (sampled softmax reference on / off)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_hip_deaot_ops.py tests/test_hip_deaot_engine.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -30 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for e in "RMEM_GP_SAMPLE=0" "RMEM_GP_SAMPLE=1" "RMEM_GP_SAMPLE=0" "RMEM_GP_SAMPLE=1"; do
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$e $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])")"
done
