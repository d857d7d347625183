#!/bin/bash
# the other BASELINE workloads on the current tree (bench defaults), then a kernel trace of the headline workload broken down per layer shape
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w
mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -20 $O/$name.err; exit 1; }; cut -c1-400 $O/$name.json; }
run bench_cfg3_ytvos_inject --workload ytvos_720p_r50_N8_inject
run bench_cfg4_vost_N8 --workload vost_1080p_r50_N8
run bench_cfg4_vost_unbounded --workload vost_1080p_r50_unbounded
run bench_deaot --workload davis17_480p_r50deaot_N9
run bench_fp16 --dtype fp16
run bench_swin_fp16 --workload lvos_720p_swinb_N12
run bench_mixed --workload davis17_480p_r50_N8_mixed
run bench_drain --drain
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --roofline-launches 0 > $O/bench_traced.json 2> $O/trace.err || { tail -20 $O/trace.err; exit 1; }
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
head -1 $f | cut -c1-600
python scripts/trace_overlap.py $f 0.75 60 > $O/trace_per_layer.txt
cat $O/trace_per_layer.txt
rm -rf $O/tr
