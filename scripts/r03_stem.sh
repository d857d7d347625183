#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s2
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "stem" > $O/t.log 2>&1 || { grep -v "^$" $O/t.log | tail -30 | cut -c1-300; exit 1; }
tail -2 $O/t.log
timeout -k 10 200 python scripts/stem_bench.py > $O/stem_bench.txt 2>&1 || { tail -20 $O/stem_bench.txt; exit 1; }
grep frames $O/stem_bench.txt
for w in 384 1024; do echo "WGS=$w: $(RMEM_STEM_POOL_WGS=$w timeout -k 10 200 python scripts/stem_bench.py 2>&1 | grep 'one pass')"; done
timeout -k 10 600 python -m pytest tests/test_hip_engine.py -m gpu -q -x -k "bench_path or encoder or matches_per_clip" > $O/t2.log 2>&1 || { grep -v "^$" $O/t2.log | tail -30 | cut -c1-300; exit 1; }
tail -2 $O/t2.log
for env in "RMEM_STEM=direct" "X=0" "RMEM_STEM=direct" "X=0"; do
  echo "== $env"
  env $env timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 4 | cut -c1-140
done
