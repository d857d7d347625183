#!/bin/bash
# attention rewrite: op tests, micro-benchmark over groupings, engine tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "attn or attention" > gpurun_out/r2_t2.log 2>&1
rc=$?
tail -15 gpurun_out/r2_t2.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python scripts/attn_bench.py --T 8 --wgs 448,896,1792,3584 > gpurun_out/r2_attn_bench.txt 2>&1 || { tail -5 gpurun_out/r2_attn_bench.txt; exit 1; }
timeout -k 10 200 python scripts/attn_bench.py --T 1,2,3,5,7 >> gpurun_out/r2_attn_bench.txt 2>&1
cat gpurun_out/r2_attn_bench.txt
timeout -k 10 400 python -m pytest tests/test_hip_engine.py tests/test_hip_ops.py -m gpu -q -x > gpurun_out/r2_t2b.log 2>&1
rc=$?
tail -8 gpurun_out/r2_t2b.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2_b2_long.json 2> gpurun_out/r2_b2_long.err || { echo long bench failed; tail -20 gpurun_out/r2_b2_long.err; exit 1; }
cat gpurun_out/r2_b2_long.json
