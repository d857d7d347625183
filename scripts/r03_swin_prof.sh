#!/bin/bash
# kernel budget of the Swin-B workload (cfg 5) on the current tree
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o swin -- python3 bench.py --no-cpu-baseline --workload lvos_720p_swinb_N12 --steps 12 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp "$f" $O/swin_kernel_stats.csv
rm -rf $O/prof
head -30 $O/swin_kernel_stats.csv | cut -c1-170
