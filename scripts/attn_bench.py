#!/usr/bin/env python3
"""Stand-alone timing of the long-term memory read (rmem_mem_read_attn_clips) at the bench's group shape:
HW = 1674 queries, bank of T frames, 8 heads x 32, B clips per launch.  Prints us per launch and TFLOP/s (4*HW*T*HW*256 per
clip) for a few workgroup-count targets (RMEM_ATTN_WGS: how many table rows one workgroup walks) -- the knob behind
rmem_ocu_amd/csrc/attention.hip's grouping heuristic.  Usage: python scripts/attn_bench.py [--T 8] [--clips 4] [--wgs 448,1792,...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--T', default='8')
    ap.add_argument('--clips', type=int, default=4)
    ap.add_argument('--L', type=int, default=1674)
    ap.add_argument('--wgs', default='0')
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--mass', action='store_true')
    args = ap.parse_args()
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.runtime import temporal_slots
    dev = torch.device('cuda', 0)
    B, L, C = args.clips, args.L, 256
    g = torch.Generator().manual_seed(1)
    for T in [int(t) for t in args.T.split(',')]:
        S = T + 1
        q = torch.randn(B, L, C, generator=g).to(torch.bfloat16).to(dev)
        kb = torch.randn(B * S, L, C, generator=g).to(torch.bfloat16).to(dev)
        vb = torch.randn(B * S, L, C, generator=g).to(torch.bfloat16).to(dev)
        pe_cur = torch.randn(C, generator=g).to(dev)
        pe_mem = torch.randn(4, C, generator=g).to(dev)
        slots = temporal_slots(T)
        splits = max(1, min(8 // T, 32 // T))
        per = (L + splits - 1) // splits
        rows = [(c * S + t, j * per, min(per, L - j * per), slots[t], t) for c in range(B) for t in range(T) for j in range(splits)]
        n = T * splits
        tab = ops.make_chunk_table(rows).to(dev)
        out = torch.zeros(B, L, C, dtype=torch.bfloat16, device=dev)
        mass = torch.zeros(B, L, T, dtype=torch.float32, device=dev) if args.mass else None
        ws = ops.attn_workspace(L, 8, n, dev, nclips=B)
        flops = 4.0 * L * T * L * C * B
        for w in args.wgs.split(','):
            if int(w) > 0:
                os.environ['RMEM_ATTN_WGS'] = w
            else:
                os.environ.pop('RMEM_ATTN_WGS', None)
            op = ops.mem_read_attn(q, kb, vb, out, ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=L * C, chunks=tab, nchunks=n,
                                   pe_cur=pe_cur, pe_mem=pe_mem, mass=mass, T=T, nclips=B, q_cs=L * C, out_cs=L * C)
            for _ in range(5):
                ops.run(op)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.run(op)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            med = ts[len(ts) // 2]
            print(f'T={T} clips={B} rows={n} target_wgs={w:>8s}: median {med:8.1f} us  min {ts[0]:8.1f} us  '
                  f'{flops / med / 1e6:7.1f} TFLOP/s (all launches of the call: attention + combine + mass)', flush=True)
            from rmem_ocu_amd import _lib
            L = _lib.lib()
            if hasattr(L, 'rmem_attn_timeline_read'):       # measurement build (scripts/build_attn_variants.sh timeline=...)
                import ctypes
                buf = (ctypes.c_double * 5)()
                L.rmem_attn_timeline_read.restype = ctypes.c_int
                rc = L.rmem_attn_timeline_read(buf, 1792)
                print(f'  timeline (cycles per tile, wave 0 of each workgroup; rc {rc}): LDS-K wait {buf[0]:.0f}  QK->first exp {buf[1]:.0f}  '
                      f'exp/PV/sum {buf[2]:.0f}  scalar+DMA+barrier {buf[3]:.0f}  total {sum(buf[:4]):.0f}   tiles/WG {buf[4]:.1f}', flush=True)


if __name__ == '__main__':
    main()
