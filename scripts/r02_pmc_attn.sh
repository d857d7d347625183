#!/bin/bash
# PMC passes over the memory-read kernel at the bench's group shape (T = 8, 4 clips): rocprofv3 --pmc, one counter set per run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run() {  # name, counters
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc/$1 -o p -- python3 scripts/attn_bench.py --T 8 --iters 10 > gpurun_out/pmc/$1.log 2>&1 || { echo "pass $1 failed"; tail -5 gpurun_out/pmc/$1.log; return 1; }
  f=$(find gpurun_out/pmc/$1 -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$1" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'k_attn_partial' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v.sort()
    print(f'{sys.argv[2]:10s} {k:32s} n={len(v):3d} median={v[len(v)//2]:.4g}')
PY
  rm -rf gpurun_out/pmc/$1
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" &&
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" &&
run sq3 "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES_EQ_64 SQ_INSTS_FLAT" &&
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT" &&
run fetch "FETCH_SIZE" &&
run write "WRITE_SIZE"
