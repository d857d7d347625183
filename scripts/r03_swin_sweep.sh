#!/bin/bash
# groups in flight: Swin-B default (one group of 8) and the DeAOT workload at 8 / 16 / 24 clips in flight
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
run() { w=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w "$@" > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$w $* $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['clips_in_flight_per_gpu'])")"; }
run lvos_720p_swinb_N12 &&
cp $O/c.txt $O/bench_swin_fp16.json &&
run davis17_480p_r50deaot_N9 --clips-in-flight 8 &&
run davis17_480p_r50deaot_N9 --clips-in-flight 16 &&
run davis17_480p_r50deaot_N9 &&
run davis17_480p_r50_N8 --clips-in-flight 16 &&
run davis17_480p_r50_N8
