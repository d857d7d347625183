#!/usr/bin/env python3
"""rmem_bneck_chain against the two launches it replaces (conv3 + residual, then the next conv1) at the layer-1 geometry of the
bench: --images frames of 121 x 213 pixels, 64 -> 256 -> N2 channels.  --reps launches back to back per event pair, rotating over
--sets operand sets (the 256-channel maps are 211 MB each at 16 images: nothing is found in a cache).
Usage: python scripts/bneck_bench.py [--images 16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=16)
    ap.add_argument('--reps', type=int, default=9)
    ap.add_argument('--sets', type=int, default=3)
    ap.add_argument('--iters', type=int, default=7)
    args = ap.parse_args()
    from rmem_ocu_amd import ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    B, H, W = args.images, 121, 213
    M = B * H * W
    bf = torch.bfloat16
    r = lambda *s, sc=0.5: (torch.randn(*s, generator=g) * sc).to(bf).to(dev)   # noqa: E731
    w3, b3 = r(256, 64, sc=0.1), torch.randn(256, generator=g).to(dev)
    for N2 in (64, 128):
        w1, b1 = r(N2, 256, sc=0.05), torch.randn(N2, generator=g).to(dev)
        b0, res0 = r(M, 64), r(M, 256)
        unf, fus = [], []
        for _ in range(args.sets):
            b, res = b0.clone(), res0.clone()
            y = torch.empty(M, 256, dtype=bf, device=dev)
            a2 = torch.empty(M, N2, dtype=bf, device=dev)
            unf.append([ops.conv2d(b, w3, b3, y, H=H, W=W, Cin=64, Cout=256, residual=res, relu=True, batch=B),
                        ops.conv2d(y, w1, b1, a2, H=H, W=W, Cin=256, Cout=N2, relu=True, batch=B)])
            fus.append([ops.bneck_chain(b, w3, b3, y, w1, b1, a2, H=H, W=W, K1=64, N2=N2, residual=res, batch=B)])
        for name, sets in (('conv3 + residual, conv1 (two launches)', unf), ('rmem_bneck_chain', fus)):
            for s in sets:
                ops.run(s)
            torch.cuda.synchronize()
            ts = []
            for _ in range(args.iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for k in range(args.reps):
                    ops.run(sets[k % args.sets])
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / args.reps)
            ts.sort()
            mb = M * (64 + 256 + 256 + N2) * 2 / 1e6 + (0 if 'chain' in name else M * 256 * 2 / 1e6)
            t = ts[len(ts) // 2]
            print(f'{B} images, N2 = {N2}: {name:42s} {t:7.1f} us  ({mb:.0f} MB of compulsory traffic: {mb / t:.2f} TB/s)', flush=True)


if __name__ == '__main__':
    main()
