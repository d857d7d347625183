#!/bin/bash
# the headline line on the last tree of the round
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final3
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > gpurun_out/final3/bench_driver_form.json 2> gpurun_out/final3/err.txt || { tail -5 gpurun_out/final3/err.txt; exit 1; }
cut -c1-170 gpurun_out/final3/bench_driver_form.json
