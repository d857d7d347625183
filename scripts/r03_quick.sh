#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scripts/idbank_bench.py 2>&1 | grep "id_embed"
RMEM_IDB_PANEL_MAJOR=1 timeout -k 10 200 python scripts/idbank_bench.py 2>&1 | grep "id_embed"
timeout -k 10 200 python scripts/idbank_bench.py 2>&1 | grep "id_embed"
RMEM_IDB_PANEL_MAJOR=1 timeout -k 10 200 python scripts/idbank_bench.py 2>&1 | grep "id_embed"
