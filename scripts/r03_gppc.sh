#!/bin/bash
# producer / consumer form of the gated attention's P.V kernel: DeAOT op tests under the switch, then workload A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
mkdir -p $O
RMEM_GP_PC=1 timeout -k 10 400 python -m pytest tests/test_hip_deaot_ops.py -x -q -m gpu > $O/t.txt 2>&1 || { tail -20 $O/t.txt; exit 1; }
tail -2 $O/t.txt
for e in "RMEM_GP_PC=0" "RMEM_GP_PC=1" "RMEM_GP_PC=0" "RMEM_GP_PC=1"; do
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline --workload davis17_480p_r50deaot_N9 > $O/c.txt 2>&1 || { tail -5 $O/c.txt; exit 1; }
  echo "$e $(python -c "import json,sys; d=json.loads(open('$O/c.txt').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('roofline'))")"
done
