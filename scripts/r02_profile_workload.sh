#!/bin/bash
# per-kernel time split of one bench workload: bash scripts/r02_profile_workload.sh <workload> [extra bench args]
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
W=$1; shift
rm -rf gpurun_out/prof_$W
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$W -o p -- python3 bench.py --workload $W --no-cpu-baseline \
  --roofline-launches 0 "$@" > gpurun_out/bench_${W}_under_rocprof.json 2> gpurun_out/prof_$W.err || { tail -20 gpurun_out/prof_$W.err; exit 1; }
f=$(find gpurun_out/prof_$W -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${W}_kernel_stats.csv
rm -rf gpurun_out/prof_$W
cat gpurun_out/bench_${W}_under_rocprof.json | cut -c1-300
head -28 gpurun_out/${W}_kernel_stats.csv | cut -c1-170
