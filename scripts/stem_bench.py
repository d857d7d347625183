#!/usr/bin/env python3
"""The ResNet stem (7x7 stride 2, 3 -> 64) at the bench geometry, --images frames of 481 x 849: rmem_stem7x7s2 on the zero-bordered NHWC4
layout against the generic row-run GEMM form on the 8-channel layout, each with its layout kernel.  --reps launches back to back per
event pair, rotating over --sets operand sets.  Usage: python scripts/stem_bench.py [--images 16]"""
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
    B, H, W = args.images, 481, 849
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    bf = torch.bfloat16
    w = (torch.randn(64, 7, 7, 3, generator=g) * 0.08).to(bf)
    w8 = torch.nn.functional.pad(w, (0, 5)).contiguous().to(dev)
    w4 = torch.zeros(64, 8, 8, 4, dtype=bf)
    w4[:, :7, :7, :3] = w
    w4 = w4.to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    hp, wp = ops.stem_padded_size(H, W)
    new, old, conv_new, conv_old, fused, two = [], [], [], [], [], []
    for _ in range(args.sets):
        imgs = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(B)]
        ptrs = torch.tensor([i.data_ptr() for i in imgs], dtype=torch.int64, device=dev)
        x4 = torch.zeros(B, hp, wp, 4, dtype=bf, device=dev)
        x8 = torch.zeros(B, H * W, 8, dtype=bf, device=dev)
        y = torch.empty(B, Ho * Wo, 64, dtype=bf, device=dev)
        a = [ops.image_ptrs_to_nhwc4p(ptrs, x4, H=H, W=W, images=B), ops.stem7x7s2(x4, w4, bias, y, H=H, W=W, images=B)]
        b = [ops.image_ptrs_to_nhwc8(ptrs, x8, H=H, W=W, images=B),
             ops.conv2d(x8, w8, bias, y, H=H, W=W, Cin=8, Cout=64, KH=7, KW=7, stride=2, pad=3, relu=True, batch=B)]
        Hq, Wq = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
        yp = torch.empty(B, Hq * Wq, 64, dtype=bf, device=dev)
        fused.append([ops.stem7x7s2_pool(x4, w4, bias, yp, H=H, W=W, images=B)])
        two.append([a[1], ops.maxpool3x3s2(y, yp, H=Ho, W=Wo, C=64, images=B)])
        new.append(a); old.append(b); conv_new.append(a[1:]); conv_old.append(b[1:])
        new[-1].append(imgs)            # keep the frames alive
    for name, sets in (('layout + row-run GEMM (8 channels, K = 448)', old), ('  its GEMM alone', conv_old),
                       ('layout + rmem_stem7x7s2 (4 channels, K = 256)', new), ('  rmem_stem7x7s2 alone', conv_new),
                       ('rmem_stem7x7s2 + rmem_maxpool3x3s2', two), ('rmem_stem7x7s2_pool (one pass)', fused)):
        sets = [[o for o in s if not isinstance(o, list)] for s in sets]
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
        print(f'{B} frames: {name:48s} {ts[len(ts) // 2]:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
