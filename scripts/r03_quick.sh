#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_hip_ops.py -m gpu -q -x -k "label_id_embed" 2>&1 | tail -2
timeout -k 10 200 python scripts/idbank_bench.py 2>&1 | grep clips
