#!/bin/bash
# after the late DeAOT changes (sampled softmax reference): GPU tier on the final tree, the DeAOT workload plain and under rocprofv3,
# and the headline line once more (the AOT path does not touch gated_attn.hip; this shows it unchanged)
cd $GRAFT_REPO_ROOT
O=gpurun_out/final2
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
rc=$?
tail -3 $O/gpu_tests.log
[ $rc -eq 0 ] || { echo "GPU tier failed"; exit $rc; }
timeout -k 10 300 python bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 > $O/bench_deaot.json 2> $O/err.txt || { tail -5 $O/err.txt; exit 1; }
cut -c1-160 $O/bench_deaot.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o deaot -- python3 bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 --steps 20 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$f" $O/deaot_kernel_stats.csv; rm -rf $O/prof
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/err2.txt || { tail -5 $O/err2.txt; exit 1; }
cut -c1-160 $O/bench_driver_form.json
