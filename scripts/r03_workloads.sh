#!/bin/bash
# the other BASELINE workloads on the current tree (bench defaults); cfg 4 geometry also as whole drained jobs (600-frame clips)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 500 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -20 $O/$name.err; exit 1; }; cut -c1-200 $O/$name.json; }
run bench_cfg3_ytvos_inject --workload ytvos_720p_r50_N8_inject
run bench_cfg4_vost_N8 --workload vost_1080p_r50_N8
run bench_cfg4_vost_N8_drained --workload vost_1080p_r50_N8 --drain
run bench_cfg4_vost_unbounded_drained --workload vost_1080p_r50_unbounded --drain
run bench_deaot --workload davis17_480p_r50deaot_N9
run bench_fp16 --dtype fp16
run bench_swin_fp16 --workload lvos_720p_swinb_N12
run bench_mixed --workload davis17_480p_r50_N8_mixed
run bench_drain --drain
