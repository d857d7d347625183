#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err || { tail -20 gpurun_out/bench_check.err; exit 1; }
python -c "
import json; d=json.loads(open('gpurun_out/bench_check.json').read().strip().splitlines()[-1]); print(d['value'], d['check'], d['cpu_baseline'])"
