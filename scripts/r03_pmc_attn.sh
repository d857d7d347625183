#!/bin/bash
# PMC passes over the memory-read kernel at the bench's group shape (T = 8, 8 clips per launch): rocprofv3 --pmc, one counter set per run
# (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, rocprofv3 PMC slots); prints medians, writes the JSON
# bench.py's `roofline.traffic` reads (it names the sha256 of the attention.hip it was measured on)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
run() {  # name, counters
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc/$1 -o p -- python3 scripts/attn_bench.py --T 8 --clips 8 --iters 10 > gpurun_out/pmc/$1.log 2>&1 || { echo "pass $1 failed"; tail -5 gpurun_out/pmc/$1.log; return 1; }
  f=$(find gpurun_out/pmc/$1 -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$1" <<'PY'
import csv, sys, collections, json, os
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'k_attn_partial' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
med = {}
for k, v in acc.items():
    v.sort()
    med[k] = v[len(v) // 2]
    print(f'{sys.argv[2]:10s} {k:32s} n={len(v):3d} median={v[len(v)//2]:.4g}')
json.dump(med, open(f'gpurun_out/pmc/{sys.argv[2]}.json', 'w'))
PY
  rm -rf gpurun_out/pmc/$1
}
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" &&
run sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" &&
run grbm "GRBM_GUI_ACTIVE GRBM_COUNT" &&
run fetch "FETCH_SIZE" &&
run write "WRITE_SIZE" &&
python3 - <<'PY'
import json, hashlib
f = json.load(open('gpurun_out/pmc/fetch.json'))['FETCH_SIZE']
w = json.load(open('gpurun_out/pmc/write.json'))['WRITE_SIZE']
sha = hashlib.sha256(open('rmem_ocu_amd/csrc/attention.hip', 'rb').read()).hexdigest()
alg = 8 * (2 * 8 * 1674 * 256 * 2 + 2 * 1674 * 256 * 2)
hbm = int((2 * f + w) * 1024)
out = {'kernel': 'k_attn_partial<true, *> (memory read, HW=1674, T=8, 8 heads, 8 clips per launch: rmem_mem_read_attn_clips; 1792 workgroups walking 4 bank frames each)',
       'command': 'bash scripts/r03_pmc_attn.sh  (rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 scripts/attn_bench.py --T 8 --clips 8 --iters 10, and a second pass with --pmc WRITE_SIZE); medians over the launches',
       'FETCH_SIZE_KB_median': f, 'WRITE_SIZE_KB_median': w,
       'correction': 'gfx950 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact',
       'hbm_bytes_per_launch': hbm, 'algorithmic_bytes_per_launch': alg, 'ratio': round(hbm / alg, 3), 'attention_hip_sha256': sha}
json.dump(out, open('gpurun_out/pmc/attn_pmc_group8.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
