#!/bin/bash
# scratch: one short bench line of the current tree (used for A/B runs of experiment switches during the round)
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 4 "$@" | cut -c1-200
