#!/usr/bin/env python3
"""The id bank of a label map (8 clips at 480 x 854 -> 481 x 849, 17 x 17 stride 16): rmem_label_id_embed (one-hot operand built in
registers) against rmem_label_to_onehot16 + rmem_conv2d_nhwc.  Usage: python scripts/idbank_bench.py [--clips 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=8)
    args = ap.parse_args()
    from rmem_ocu_amd import ops
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(0)
    B, hs, ws, H, W, K = args.clips, 480, 854, 481, 849, 17
    bf = torch.bfloat16
    w = (torch.randn(256, K, K, 16, generator=g) * 0.05).to(bf).to(dev)
    bias = torch.randn(256, generator=g).to(dev)
    lab = torch.randint(0, 11, (B, hs, ws), generator=g).to(torch.uint8).to(dev)
    Ho, Wo = (H + 16 - K) // 16 + 1, (W + 16 - K) // 16 + 1
    out = torch.empty(B * Ho * Wo, 256, dtype=bf, device=dev)
    scratch = ops.label_id_embed_scratch(B, H, W, 8, dev)
    oh = torch.empty(B * H * W, 16, dtype=bf, device=dev)
    new = [ops.label_id_embed(lab, w, bias, scratch, out, Hs=hs, Ws=ws, H=H, W=W, K=K, stride=16, pad=8, images=B)]
    old = [ops.label_to_onehot16(lab, oh, Hs=hs, Ws=ws, Hd=H, Wd=W, images=B),
           ops.conv2d(oh, w, bias, out, H=H, W=W, Cin=16, Cout=256, KH=K, KW=K, stride=16, pad=8, batch=B)]
    for name, prog in (('rmem_label_to_onehot16 + rmem_conv2d_nhwc', old), ('rmem_label_id_embed', new)):
        for _ in range(3):
            ops.run(prog)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                ops.run(prog)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 8)
        ts.sort()
        print(f'{B} clips: {name:48s} {ts[3]:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
