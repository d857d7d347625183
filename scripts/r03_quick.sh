#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scripts/idbank_bench.py 2>&1 | grep "id_embed"
