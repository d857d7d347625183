#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_engine.py tests/test_hip_deaot_engine.py -m gpu -q -s -k "n2 or fp16 or swin or new_object or unbounded" > gpurun_out/r2_t9.log 2>&1
grep -v "^\.*$" gpurun_out/r2_t9.log | grep -v "amdgpu.ids" | tail -30
for d in bf16 fp16; do
timeout -k 10 300 python bench.py --no-cpu-baseline --dtype $d > gpurun_out/r2_b9_$d.json 2> gpurun_out/r2_b9_$d.err || { echo bench $d failed; tail -20 gpurun_out/r2_b9_$d.err; exit 1; }
cat gpurun_out/r2_b9_$d.json
done
timeout -k 10 400 python bench.py --no-cpu-baseline --workload lvos_720p_swinb_N12 --steps 300 --warmup 30 > gpurun_out/r2_b9_swin_fp16.json 2> gpurun_out/r2_b9_swin.err || { echo swin bench failed; tail -20 gpurun_out/r2_b9_swin.err; exit 1; }
cat gpurun_out/r2_b9_swin_fp16.json
