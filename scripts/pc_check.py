#!/usr/bin/env python3
"""Hashes of rmem_conv2d_nhwc outputs on the shapes that take the 128x128 tile (K >= 512) and of a gated attention (P.V kernel): run
once per RMEM_GEMM_PC / RMEM_GP_PC setting and compare the printed lines -- the producer / consumer forms must be bit-identical (same
MFMA order).  The switches are read once per process, hence a script: tests/test_hip_ops.py::test_producer_consumer_forms_are_bit_identical
runs it in child processes."""
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

# gated attention (k_gp_pv with / without loader waves): T = 3 frames of 333 keys, ragged tiles, two clips
import math
T, L, NC = 3, 333, 2
q = (torch.randn(NC, L, 128, generator=g) * 1.5).to(torch.bfloat16).to(dev)
k = (torch.randn(NC * T, L, 128, generator=g) * 1.5).to(torch.bfloat16).to(dev)
v = torch.randn(NC * T, L, 1024, generator=g).to(torch.bfloat16).to(dev)
u = torch.randn(NC, L, 1024, generator=g).to(torch.bfloat16).to(dev)
rows = [(c * T + t, 0, L, -1, t) for c in range(NC) for t in range(T)]
ws = ops.gated_workspace(L, 1024, T, L, T, dev, nclips=NC)
out = torch.zeros(NC, L, 1024, dtype=torch.bfloat16, device=dev)
ops.run(ops.gated_attn(q, k, v, u, out, ws, Lq=L, DV=1024, ldq=128, ldk=128, ldv=1024, ldua=1024, ldo=1024, k_slot_stride=L * 128,
                       v_slot_stride=L * 1024, chunks=ops.make_chunk_table(rows).to(dev), nchunks=T, frames=T, keys_per_frame=L, nclips=NC))
torch.cuda.synchronize()
assert torch.isfinite(out.float()).all() and float(out.float().abs().mean()) > 1e-3
print('gated', hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16], float(out.float().abs().mean()))
