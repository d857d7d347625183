#!/bin/bash
# full GPU tier + driver-form bench + the same under rocprofv3 + long bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r2_t6.log 2>&1
rc=$?
tail -5 gpurun_out/r2_t6.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b6.json 2> gpurun_out/r2_b6.err || { echo bench failed; tail -20 gpurun_out/r2_b6.err; exit 1; }
cat gpurun_out/r2_b6.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof6 -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b6_rocprof.json 2> gpurun_out/r2_b6_rocprof.err || { echo rocprof bench failed; tail -20 gpurun_out/r2_b6_rocprof.err; exit 1; }
cat gpurun_out/r2_b6_rocprof.json
find gpurun_out/r2_prof6 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_b6_kernel_stats.csv \;
rm -rf gpurun_out/r2_prof6
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2_b6_long.json 2> gpurun_out/r2_b6_long.err || { echo long bench failed; tail -20 gpurun_out/r2_b6_long.err; exit 1; }
cat gpurun_out/r2_b6_long.json
