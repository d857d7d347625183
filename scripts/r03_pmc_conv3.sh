#!/bin/bash
# PMC passes over one of the round-3 kernels (where do the cycles go?): usage r03_pmc_conv3.sh <bench script> <kernel name substring>
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc3
export TMPDIR=/tmp
S=${1:-scripts/idbank_bench.py}
K=${2:-k_idbank_labels}
run() {  # name, counters
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc3/$1 -o p -- python3 $S > gpurun_out/pmc3/$1.log 2>&1 || { echo "pass $1 failed"; tail -5 gpurun_out/pmc3/$1.log; return 1; }
  f=$(find gpurun_out/pmc3/$1 -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$1" "$K" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if sys.argv[3] in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v.sort()
    print(f'{sys.argv[3]:18s} {k:32s} n={len(v):3d} median={v[len(v)//2]:.4g}')
PY
  rm -rf gpurun_out/pmc3/$1
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" &&
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" &&
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT"
