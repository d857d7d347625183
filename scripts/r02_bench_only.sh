#!/bin/bash
# the bench legs of r02_final.sh alone (driver form, the same under rocprofv3 --kernel-trace --stats, long form)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err || { echo bench failed; tail -20 $O/bench_driver_form.err; exit 1; }
cat $O/bench_driver_form.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_form_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; tail -20 $O/rocprof.err; exit 1; }
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_driver_form_kernel_stats.csv \;
rm -rf $O/prof
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_long.json 2> $O/bench_long.err || { echo long bench failed; tail -20 $O/bench_long.err; exit 1; }
cat $O/bench_long.json
