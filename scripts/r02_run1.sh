#!/bin/bash
# first GPU call of round 2: tests (incl. stage budget), driver-form bench, the same under rocprofv3, long bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -q -s > gpurun_out/r2_t1.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/r2_t1.log
if [ $rc -gt 1 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_b1.json 2> gpurun_out/r2_b1.err || { echo bench failed; tail -20 gpurun_out/r2_b1.err; exit 1; }
cat gpurun_out/r2_b1.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2_prof1 -o bench -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b1_rocprof.json 2> gpurun_out/r2_b1_rocprof.err || { echo rocprof bench failed; tail -20 gpurun_out/r2_b1_rocprof.err; exit 1; }
cat gpurun_out/r2_b1_rocprof.json
find gpurun_out/r2_prof1 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r2_b1_kernel_stats.csv \;
rm -rf gpurun_out/r2_prof1
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2_b1_long.json 2> gpurun_out/r2_b1_long.err || { echo long bench failed; tail -20 gpurun_out/r2_b1_long.err; exit 1; }
cat gpurun_out/r2_b1_long.json
