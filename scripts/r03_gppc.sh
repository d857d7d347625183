#!/bin/bash
# gated attention kernels: DeAOT op + engine tests, then the workload under rocprofv3 (per-kernel durations) and plain
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_hip_deaot_ops.py tests/test_hip_deaot_engine.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -20 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])")"
done
rm -rf $O/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o deaot -- python3 bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 --steps 20 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/deaot_kernel_stats.csv
head -16 $O/deaot_kernel_stats.csv | cut -c1-150
