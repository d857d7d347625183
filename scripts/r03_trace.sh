#!/bin/bash
# kernel trace of the headline workload: overlap statistics and the in-situ duration of every (kernel, grid) = layer shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --roofline-launches 0 "$@" > $O/bench.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python scripts/trace_overlap.py $f 0.5 70 > $O/trace_per_layer.txt
cat $O/trace_per_layer.txt
rm -rf $O/tr
