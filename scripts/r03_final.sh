#!/bin/bash
# round-3 evidence run: GPU tier, driver-form bench, the same under rocprofv3, long bench, frame-unit bench (rounds 1-2 reading),
# PMC passes over the memory-read kernel.  Nothing is recorded from a tree whose GPU tier fails.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
rc=$?
tail -4 $O/gpu_tests.log
if [ $rc -ne 0 ]; then echo "GPU tier failed (rc $rc): no evidence is recorded from a tree whose tests fail"; exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err || { echo bench failed; tail -20 $O/bench_driver_form.err; exit 1; }
cat $O/bench_driver_form.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_form_under_rocprof.json 2> $O/rocprof.err || { echo rocprof bench failed; tail -20 $O/rocprof.err; exit 1; }
cat $O/bench_driver_form_under_rocprof.json
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/bench_driver_form_kernel_stats.csv \;
rm -rf $O/prof
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_long.json 2> $O/bench_long.err || { echo long bench failed; tail -20 $O/bench_long.err; exit 1; }
cat $O/bench_long.json
timeout -k 10 300 python bench.py --no-cpu-baseline --step-unit frame --steps 20 --warmup 5 > $O/bench_frame_unit_20.json 2> $O/bench_frame_unit.err || { echo frame-unit bench failed; tail -20 $O/bench_frame_unit.err; exit 1; }
cat $O/bench_frame_unit_20.json
bash scripts/r03_pmc_attn.sh > $O/attn_pmc_counters.txt 2>&1
cat $O/attn_pmc_counters.txt
cp gpurun_out/pmc/attn_pmc_group8.json $O/ 2>/dev/null
