#!/bin/bash
# bench sweeps: clips per group / clips in flight / look-ahead, long form and driver form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r2_sweep4.txt
for cfg in "4 24 2" "5 25 2" "5 20 2" "10 20 2" "10 30 2" "20 20 2" "20 40 2" "5 30 2"; do
  set -- $cfg
  v=$(timeout -k 10 300 python bench.py --no-cpu-baseline --clips-per-group $1 --clips-in-flight $2 --encoder-lookahead $3 --roofline-launches 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'])")
  d=$(timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --clips-per-group $1 --clips-in-flight $2 --encoder-lookahead $3 --roofline-launches 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'])")
  echo "clips_per_group=$1 clips_in_flight=$2 lookahead=$3: long $v frames/s, driver form $d frames/s" >> gpurun_out/r2_sweep4.txt
done
cat gpurun_out/r2_sweep4.txt
