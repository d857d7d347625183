"""Summarise a rocprofv3 --kernel-trace CSV: over the steady-state part of the run, the fraction of time at least one kernel runs,
the average number of kernels in flight, and per kernel name the mean duration (to compare with the same kernel running alone).
Usage: python scripts/trace_overlap.py kernel_trace.csv [skip_fraction]"""
import csv
import sys
from collections import defaultdict



def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('_GLOBAL__N_1', '')
    if n.startswith('void '):
        n = n[5:]
    return n.split('(')[0]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Stream_Id', 0) or 0),
                     (int(r.get('Grid_Size_X', 0) or 0), int(r.get('Grid_Size_Y', 0) or 0), int(r.get('Workgroup_Size_X', 0) or 0))))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + (t1 - t0) * skip
sel = [r for r in rows if r[0] >= lo]
ev = []
for s, e, *_ in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = 0; area = 0; cur = 0; last = ev[0][0]; hist = defaultdict(int)
for t, d in ev:
    dt = t - last
    if cur > 0:
        busy += dt
    area += cur * dt
    hist[min(cur, 8)] += dt
    cur += d; last = t
span = ev[-1][0] - ev[0][0]
print(f'window {span / 1e6:.2f} ms, {len(sel)} kernels: busy {100 * busy / span:.1f} %, mean kernels in flight {area / span:.2f}')
print('time share by kernels in flight:', {k: f'{100 * v / span:.1f}%' for k, v in sorted(hist.items())})
agg = defaultdict(lambda: [0, 0])
for s, e, n, *_ in sel:
    a = agg[short(n)[-60:]]; a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:22]:
    print(f'{n:60s} {c:6d} calls  mean {d / c / 1e3:8.1f} us  {100 * d / tot:5.1f} %')
print(f'sum of kernel durations / window = {tot / span:.2f}')

# the same per (kernel, grid): one row per layer shape, so that in-situ durations can be set beside scripts/gemm_bench.py's isolated ones
if len(sys.argv) > 3:
    per = defaultdict(lambda: [0, 0])
    for s, e, n, _, g in sel:
        a = per[(short(n)[-48:], g)]; a[0] += 1; a[1] += e - s
    print('per (kernel, grid x, grid y, workgroup):')
    for (n, g), (c, d) in sorted(per.items(), key=lambda x: -x[1][1])[:int(sys.argv[3])]:
        print(f'{n:48s} {str(g):24s} {c:6d} calls  mean {d / c / 1e3:8.1f} us  {100 * d / tot:5.1f} %')
