#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 120 python scripts/attn_debug.py 2>&1 | grep -c "max err   0.[0-3]" 
bash scripts/r02_run2.sh
