#!/bin/bash
# quick check of a change: the group-engine tests + two long benches + driver form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_engine.py -m gpu -q -x -k "group or lookahead or pinned or fixture" > $O/tests.log 2>&1
rc=$?
tail -5 $O/tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 $@ 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long', j['value'])" || { tail -20 $O/err.txt; exit 1; }
done
timeout -k 10 300 python bench.py --no-cpu-baseline --roofline-launches 0 --steps 20 --warmup 5 $@ 2> $O/err.txt | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver form', j['value'])" || { tail -20 $O/err.txt; exit 1; }
