#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_hip_deaot_ops.py tests/test_hip_deaot_engine.py -m gpu -q -x > $O/tests_deaot.log 2>&1
rc=$?
tail -8 $O/tests_deaot.log
if [ $rc -ne 0 ]; then echo "deaot tests failed rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 > $O/bench_deaot.json 2> $O/err.txt || { tail -20 $O/err.txt; exit 1; }
cat $O/bench_deaot.json
