#!/bin/bash
# DeAOT clip groups: parity of the group path against the per-clip engines, then the DeAOT workload with 1 / 2 / 4 clips per group.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_hip_deaot_engine.py::test_deaot_group_engine_matches_per_clip_engines \
  tests/test_hip_engine.py::test_group_engine_matches_per_clip_engines tests/test_hip_engine.py::test_group_engine_new_object_in_one_clip \
  -x -q -s > gpurun_out/deaot_group_tests.log 2>&1 || { tail -40 gpurun_out/deaot_group_tests.log; exit 1; }
tail -8 gpurun_out/deaot_group_tests.log
for g in 1 4 2; do
  timeout -k 10 300 python bench.py --workload davis17_480p_r50deaot_N9 --clips-per-group $g --steps 800 --warmup 80 \
    > gpurun_out/bench_deaot_g$g.json 2> gpurun_out/bench_deaot_g$g.err || { tail -20 gpurun_out/bench_deaot_g$g.err; exit 1; }
  python - <<PY
import json
d = json.loads(open('gpurun_out/bench_deaot_g$g.json').read().strip().splitlines()[-1])
print('G=$g', d['value'], d['ms_per_step'], d['config'].get('host_enqueue_ms_per_step'), d['roofline'])
PY
done
