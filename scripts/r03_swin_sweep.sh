#!/bin/bash
# Swin-B: whole drained job, one group against three
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
run() { w=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline --workload $w "$@" > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$w $* $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['clips_in_flight_per_gpu'], d['steps'], d['config']['job_clips'])")"; }
run lvos_720p_swinb_N12 --drain &&
cp $O/c.txt $O/bench_swin_fp16_drained.json &&
run lvos_720p_swinb_N12 --drain --clips-in-flight 24 &&
run lvos_720p_swinb_N12 --steps 120 &&
run lvos_720p_swinb_N12 --steps 120 --clips-in-flight 24
