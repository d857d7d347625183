#!/bin/bash
# rehearsal of the N > 1 path on ONE GPU: bench.py launches its own 2 ranks, both share the card (gloo collectives, TCPStore clip queue)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
RMEM_SHARE_GPU=1 RMEM_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 160 --warmup 16 --clips-in-flight 12 > gpurun_out/r2_n2_rehearsal.json 2> gpurun_out/r2_n2_rehearsal.err
echo "rc=$?"
cat gpurun_out/r2_n2_rehearsal.json
tail -5 gpurun_out/r2_n2_rehearsal.err
RMEM_SHARE_GPU=1 RMEM_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 160 --warmup 16 --clips-in-flight 12 --feeder static --workload davis17_480p_r50_N8_mixed > gpurun_out/r2_n2_rehearsal_mixed.json 2>> gpurun_out/r2_n2_rehearsal.err
echo "rc=$?"
cat gpurun_out/r2_n2_rehearsal_mixed.json
