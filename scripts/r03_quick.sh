#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "--clips-per-group 8 --clips-in-flight 24" "--clips-per-group 12 --clips-in-flight 24" "--clips-per-group 16 --clips-in-flight 32" "--clips-per-group 12 --clips-in-flight 36" "--clips-per-group 16 --clips-in-flight 48" "--clips-per-group 10 --clips-in-flight 30" "--clips-per-group 8 --clips-in-flight 24"; do
  echo "== $a: $(timeout -k 10 200 python bench.py --no-cpu-baseline --roofline-launches 0 $a 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])" 2>&1 | tail -1)"
done
