#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "chain or attn_pair" > $O/tests_new.log 2>&1
rc=$?
tail -5 $O/tests_new.log
if [ $rc -ne 0 ]; then echo "new tests failed rc=$rc"; exit $rc; fi
bash scripts/r03_prof1.sh ${1:-chain3} | grep -E "^[0-9]|k_chain"
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', j['value'])" || { tail -20 $O/err.txt; exit 1; }
done
timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 --steps 20 --warmup 5 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form', j['value'], j['ms_per_step'])" || { tail -20 $O/err.txt; exit 1; }
