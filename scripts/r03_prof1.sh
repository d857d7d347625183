#!/bin/bash
# per-kernel budget of ONE group running alone (isolated durations), current tree; $1 = tag, rest = env assignments
cd $GRAFT_REPO_ROOT
tag=$1; shift
O=gpurun_out/r03p
mkdir -p $O
export TMPDIR=/tmp
for v in "$@"; do export "$v"; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o g1 -- python3 bench.py --steps 200 --warmup 40 --no-cpu-baseline --clips-in-flight 4 --roofline-launches 0 > $O/bench_$tag.json 2> $O/prof_$tag.err || { echo profile failed; tail -20 $O/prof_$tag.err; exit 1; }
python3 -c "import json;print(json.load(open('$O/bench_$tag.json'))['value'])"
find $O/prof_$tag -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$tag.csv \;
rm -rf $O/prof_$tag
head -30 $O/kernel_stats_$tag.csv | cut -c1-160
