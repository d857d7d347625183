#!/usr/bin/env python3
"""Per-layer timing of rmem_conv2d_nhwc over the conv / linear shapes of the cfg-2 path (--images frames per encoder launch, --clips
clips per LSTT / decoder launch; round 3 bench default: 16 / 8) and of Swin-B stage 3 at 720p.  Each measurement is --reps launches
back to back inside one event pair, rotating over --sets operand sets so that a layer does not find its own input in the caches.  The tile choice is read from the environment once per process
(RMEM_GEMM_BIG256 = 256x128 tiles from that many tiles on, RMEM_GEMM_BIG, RMEM_GEMM_TILE ...): run it once per setting.
Usage: python scripts/gemm_bench.py [--iters 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (H, W, Cin, Cout, k, stride, batch): batch 8 = encoder images (scaled by --images / 8), batch 4 = clips (scaled by --clips / 4)
SHAPES = [
    (121, 213, 64, 64, 1, 1, 8), (121, 213, 64, 64, 3, 1, 8), (121, 213, 64, 256, 1, 1, 8), (121, 213, 256, 64, 1, 1, 8),
    (121, 213, 256, 128, 1, 1, 8), (121, 213, 128, 128, 3, 2, 8), (61, 107, 128, 512, 1, 1, 8), (61, 107, 512, 128, 1, 1, 8),
    (61, 107, 128, 128, 3, 1, 8), (61, 107, 512, 256, 1, 1, 8), (61, 107, 256, 256, 3, 2, 8), (31, 54, 256, 1024, 1, 1, 8),
    (31, 54, 1024, 256, 1, 1, 8), (31, 54, 256, 256, 3, 1, 8),
    (6696, 1, 1024, 256, 1, 1, 1), (6696, 1, 256, 768, 1, 1, 1), (6696, 1, 256, 256, 1, 1, 1), (6696, 1, 256, 1024, 1, 1, 1),
    (6696, 1, 1024, 256, 1, 1, 1), (31, 54, 256, 256, 3, 1, 4), (61, 107, 256, 128, 3, 1, 4), (121, 213, 128, 128, 3, 1, 4),
    (14400, 1, 512, 1536, 1, 1, 1), (14400, 1, 512, 512, 1, 1, 1), (14400, 1, 512, 2048, 1, 1, 1), (14400, 1, 2048, 512, 1, 1, 1),
    (3600, 1, 512, 1536, 1, 1, 1), (3600, 1, 2048, 512, 1, 1, 1),
    (481, 849, 8, 64, 7, 2, 8), (481, 849, 16, 256, 17, 16, 4),          # stem, id bank (row-run forms)
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=7)
    ap.add_argument('--reps', type=int, default=12)
    ap.add_argument('--sets', type=int, default=3)
    ap.add_argument('--images', type=int, default=16)
    ap.add_argument('--clips', type=int, default=8)
    ap.add_argument('--no-swin', action='store_true')
    ap.add_argument('--only-rowrun', action='store_true')
    args = ap.parse_args()
    from rmem_ocu_amd import ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    tot = 0.0
    for (H, W, ci, co, k, st, b) in SHAPES:
        if H in (14400, 3600) and args.no_swin:
            continue
        if args.only_rowrun and H != 481:
            continue
        if b == 8:
            b = args.images
        elif b == 4:
            b = args.clips
        elif H == 6696:
            H = 1674 * args.clips
        pad = k // 2
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        w = (torch.randn(co, k * k * ci, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        bias = torch.randn(co, generator=g).to(dev)
        ws = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=dev)
        x0 = (torch.randn(b * H * W, ci, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        sets = []
        for _ in range(args.sets):
            x = x0.clone()
            y = torch.empty(b * Ho * Wo, co, dtype=torch.bfloat16, device=dev)
            sets.append(ops.conv2d(x, w, bias, y, H=H, W=W, Cin=ci, Cout=co, KH=k, KW=k, stride=st, pad=pad, relu=True, batch=b, ws=ws))
        for op in sets:
            ops.run(op)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(args.reps):
                ops.run(sets[r % args.sets])
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / args.reps)
        ts.sort()
        med = ts[len(ts) // 2]
        flops = 2.0 * b * Ho * Wo * co * k * k * ci
        tot += med
        print(f'conv {H}x{W} Cin{ci} Cout{co} k{k} s{st} b{b}: {med:7.1f} us  {flops / med / 1e6:6.0f} TFLOP/s', flush=True)
    print(f'total {tot:.1f} us')


if __name__ == '__main__':
    main()
