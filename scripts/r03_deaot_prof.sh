#!/bin/bash
# kernel budget of the DeAOT workload (clip groups) on the current tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 > $O/bench_deaot.json 2> $O/bench_deaot.err || { tail -20 $O/bench_deaot.err; exit 1; }
cut -c1-200 $O/bench_deaot.json
cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o deaot -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 --steps 20 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$O/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/deaot_kernel_stats.csv
head -40 $O/deaot_kernel_stats.csv | cut -c1-160
