#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03sp
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_engine.py -m gpu -q -x -k "bench_path or encoder or matches_per_clip or lookahead" > $O/t2.log 2>&1 || { grep -v "^$" $O/t2.log | tail -30 | cut -c1-300; exit 1; }
tail -2 $O/t2.log
RMEM_ENC_FRONT_SPLIT=2 timeout -k 10 600 python -m pytest tests/test_hip_engine.py -m gpu -q -x -k "bench_path" > $O/t3.log 2>&1 || { grep -v "^$" $O/t3.log | tail -30 | cut -c1-300; exit 1; }
tail -2 $O/t3.log
for a in "X=0" "RMEM_ENC_FRONT_SPLIT=2" "RMEM_ENC_FRONT_SPLIT=4" "X=0" "RMEM_ENC_FRONT_SPLIT=2" "RMEM_ENC_FRONT_SPLIT=8"; do
  echo "== $a: $(env $a timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>&1 | tail -1)"
done
