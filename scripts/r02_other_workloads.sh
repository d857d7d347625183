#!/bin/bash
# bench lines of the other workloads (DeAOT with / without clip groups, Swin-B cfg 5 with look-ahead and with clip groups)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/other
O=gpurun_out/other
run() {  # name, args...
  n=$1; shift
  timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/$n.json 2> $O/$n.err || { echo "$n failed"; tail -20 $O/$n.err; return 1; }
  python3 - "$O/$n.json" "$n" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['value'], 'frames/s', d['dtype'], 'groups of', d['config']['clips_per_group'], 'look-ahead', d['config']['encoder_lookahead'],
      'frac', d['roofline']['frac'])
PY
}
run bench_deaot_g1 --workload davis17_480p_r50deaot_N9 --clips-per-group 1 &&
run bench_deaot_g4 --workload davis17_480p_r50deaot_N9 --clips-per-group 4 &&
run bench_swin_fp16_per_frame_encoder --workload lvos_720p_swinb_N12 --clips-per-group 1 --encoder-lookahead 1 --steps 240 --warmup 40 &&
run bench_swin_fp16_lookahead --workload lvos_720p_swinb_N12 --clips-per-group 1 --steps 240 --warmup 40 &&
run bench_swin_fp16_groups --workload lvos_720p_swinb_N12 --clips-per-group 4 --steps 240 --warmup 40 &&
run bench_fp16 --dtype fp16
