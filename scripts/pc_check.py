#!/usr/bin/env python3
"""Hashes of rmem_conv2d_nhwc outputs on the shapes that take the 128x128 tile (K >= 512): run once per RMEM_GEMM_PC setting and
compare the printed lines -- the producer / consumer form must be bit-identical (same MFMA order)."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_ocu_amd import ops

dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(1)
for (B, H, W, ci, co, k, st, relu) in [(16, 61, 107, 128, 128, 3, 1, True), (16, 61, 107, 512, 128, 1, 1, True), (16, 31, 54, 1024, 256, 1, 1, False),
                                       (16, 31, 54, 256, 256, 3, 1, True), (16, 61, 107, 256, 256, 3, 2, True), (3, 33, 47, 512, 384, 1, 1, False),
                                       (8, 121, 213, 128, 128, 3, 1, True), (1, 1674 * 8, 1, 1024, 256, 1, 1, False)]:
    x = (torch.randn(B, H, W, ci, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(co, k * k * ci, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    b = torch.randn(co, generator=g).to(dev)
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    y = torch.empty(B * Ho * Wo, co, dtype=torch.bfloat16, device=dev)
    ops.run(ops.conv2d(x, w, b, y, H=H, W=W, Cin=ci, Cout=co, KH=k, KW=k, stride=st, pad=pad, relu=relu, batch=B))
    torch.cuda.synchronize()
    h = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
    print(B, H, W, ci, co, k, st, tuple(y.shape), h, float(y.float().abs().mean()))
