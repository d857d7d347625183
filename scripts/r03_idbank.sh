#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03id
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "label_id_embed or onehot" > $O/t.log 2>&1 || { grep -v "^$" $O/t.log | tail -30 | cut -c1-300; exit 1; }
tail -2 $O/t.log
timeout -k 10 200 python scripts/idbank_bench.py 2>&1 | grep clips | tee $O/idbank_bench.txt
timeout -k 10 600 python -m pytest tests/test_hip_engine.py -m gpu -q -x -k "bench_path or matches_per_clip or new_object_in_one" > $O/t2.log 2>&1 || { grep -v "^$" $O/t2.log | tail -30 | cut -c1-300; exit 1; }
tail -2 $O/t2.log
for env in "RMEM_NO_LABEL_EMBED=1" "X=0" "RMEM_NO_LABEL_EMBED=1" "X=0"; do
  echo "== $env"
  env $env timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 4 | cut -c1-140
done
