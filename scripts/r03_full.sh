#!/bin/bash
# GPU tier + the cfg-4 workloads (restricted vs unbounded bank), drained
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
rc=$?
tail -12 $O/gpu_tests.log
grep -E "evictions|group clip|chain statistics" $O/gpu_tests.log | head -20
if [ $rc -ne 0 ]; then echo "GPU tier failed rc=$rc"; exit $rc; fi
for wl in vost_1080p_r50_N8 vost_1080p_r50_unbounded; do
  timeout -k 10 600 python bench.py --no-cpu-baseline --workload $wl --drain > $O/bench_$wl.json 2> $O/bench_$wl.err || { echo $wl failed; tail -20 $O/bench_$wl.err; exit 1; }
  cat $O/bench_$wl.json
done
