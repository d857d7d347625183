#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/r02_pmc_attn.sh > gpurun_out/r2_pmc_attn.txt 2>&1
cat gpurun_out/r2_pmc_attn.txt
timeout -k 10 500 python -m pytest tests/test_hip_engine.py tests/test_hip_ops.py -m gpu -q > gpurun_out/r2_t4.log 2>&1
tail -8 gpurun_out/r2_t4.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2_b4_long.json 2> gpurun_out/r2_b4_long.err || { echo long bench failed; tail -20 gpurun_out/r2_b4_long.err; exit 1; }
cat gpurun_out/r2_b4_long.json
