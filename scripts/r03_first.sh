#!/bin/bash
# round-3 first GPU call: GPU tier (incl. the new bench-geometry group test), driver-form bench, the new cfg-3 / cfg-4 workloads,
# --drain, and a per-kernel budget of ONE group running alone (what the fusion work is planned against)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
rc=$?
tail -15 $O/gpu_tests.log
if [ $rc -ne 0 ]; then echo "GPU tier failed rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err || { echo bench failed; tail -20 $O/bench_driver_form.err; exit 1; }
cat $O/bench_driver_form.json
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_long.json 2> $O/bench_long.err || { echo long bench failed; tail -20 $O/bench_long.err; exit 1; }
cat $O/bench_long.json
timeout -k 10 300 python bench.py --no-cpu-baseline --drain > $O/bench_drain.json 2> $O/bench_drain.err || { echo drain bench failed; tail -20 $O/bench_drain.err; exit 1; }
cat $O/bench_drain.json
timeout -k 10 300 python bench.py --no-cpu-baseline --workload ytvos_720p_r50_N8_inject --steps 700 --warmup 70 > $O/bench_cfg3.json 2> $O/bench_cfg3.err || { echo cfg3 bench failed; tail -20 $O/bench_cfg3.err; exit 1; }
cat $O/bench_cfg3.json
# one group alone on the GPU: per-kernel budget (isolated durations)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o g1 -- python3 bench.py --steps 400 --warmup 40 --no-cpu-baseline --clips-in-flight 4 > $O/bench_single_group.json 2> $O/prof1.err || { echo single-group profile failed; tail -20 $O/prof1.err; exit 1; }
cat $O/bench_single_group.json
find $O/prof1 -name "*kernel_stats.csv" -exec cp {} $O/single_group_kernel_stats.csv \;
rm -rf $O/prof1
