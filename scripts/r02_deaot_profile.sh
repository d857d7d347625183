#!/bin/bash
# DeAOT workload: host cost per step without back-pressure (short runs), then the per-kernel time split of the clip-group path.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
G=${1:-4}
for g in ${SHORT:-1 $G}; do
  timeout -k 10 300 python bench.py --workload davis17_480p_r50deaot_N9 --clips-per-group $g --steps 24 --warmup 8 --roofline-launches 0 \
    > gpurun_out/bench_deaot_short_g$g.json 2> gpurun_out/bench_deaot_short_g$g.err || { tail -20 gpurun_out/bench_deaot_short_g$g.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/bench_deaot_short_g$g.json').read().strip().splitlines()[-1])
print('short G=$g', d['value'], d['ms_per_step'], 'host', d['config'].get('host_enqueue_ms_per_step'))
PY
done
export TMPDIR=/tmp
rm -rf gpurun_out/prof_deaot
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_deaot -o deaot -- python3 bench.py --workload davis17_480p_r50deaot_N9 \
  --clips-per-group $G --steps 400 --warmup 40 --roofline-launches 0 > gpurun_out/bench_deaot_prof_g$G.json 2> gpurun_out/bench_deaot_prof.err \
  || { tail -20 gpurun_out/bench_deaot_prof.err; exit 1; }
f=$(find gpurun_out/prof_deaot -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/deaot_g${G}_kernel_stats.csv
find gpurun_out/prof_deaot -type f ! -name '*kernel_stats.csv' -delete
head -32 gpurun_out/deaot_g${G}_kernel_stats.csv | cut -c1-150
