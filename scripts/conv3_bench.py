#!/usr/bin/env python3
"""The C -> C 3x3 convs of the path (ResNet layer 1: 64 channels at 121 x 213; layer 2: 128 at 61 x 107; decoder conv_4x: 128 at 121 x 213
for --images / 2 clips): rmem_conv3x3_direct against the implicit-GEMM form of rmem_conv2d_nhwc.  --reps launches back to back per event pair, rotating over --sets operand sets.
Usage: python scripts/conv3_bench.py [--images 16]"""
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
    bf = torch.bfloat16
    for (B, H, W, C, what) in ((args.images, 121, 213, 64, 'layer 1'), (args.images, 61, 107, 128, 'layer 2'), (args.images // 2, 121, 213, 128, 'decoder conv_4x')):
        w = (torch.randn(C, 3, 3, C, generator=g) * 0.04).to(bf).to(dev)
        bias = torch.randn(C, generator=g).to(dev)
        x0 = (torch.randn(B * H * W, C, generator=g) * 0.5).to(bf).to(dev)
        old, new = [], []
        for _ in range(args.sets):
            x = x0.clone()
            y = torch.empty(B * H * W, C, dtype=bf, device=dev)
            old.append([ops.conv2d(x, w, bias, y, H=H, W=W, Cin=C, Cout=C, KH=3, KW=3, stride=1, pad=1, relu=True, batch=B)])
            new.append([ops.conv3x3_direct(x, w, bias, y, H=H, W=W, C=C, images=B, relu=True)])
        for name, sets in (('rmem_conv2d_nhwc (implicit GEMM)', old), ('rmem_conv3x3_direct', new)):
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
            t = ts[len(ts) // 2]
            print(f'{what} ({B} x {H} x {W} x {C}): {name:36s} {t:7.1f} us  {2.0 * B * H * W * C * 9 * C / t / 1e6:6.0f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
