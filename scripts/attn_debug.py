#!/usr/bin/env python3
"""Debug aid: attention through rmem_mem_read_attn in controlled cases; prints where the result departs from the expectation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rmem_ocu_amd import ops  # noqa: E402

dev = torch.device('cuda', 0)
BF16 = torch.bfloat16
C = 256


def ref_attn(q, k, v):
    L = q.shape[0]
    Qh = (q / 32 ** 0.5).reshape(L, 8, 32).permute(1, 0, 2)
    return (torch.softmax(Qh @ k.reshape(-1, 8, 32).permute(1, 2, 0), -1) @ v.reshape(-1, 8, 32).permute(1, 0, 2)).permute(1, 0, 2).reshape(L, C)


def launch(q, k, v, mode, mass_on):
    """mode 'plain': chunks None; 'table': the same keys described by a one-row table (memory-read instance, no PE)."""
    L, Lk = q.shape[0], k.shape[0]
    out = torch.zeros(L, C, dtype=BF16, device=dev)
    ws = ops.attn_workspace(L, 8, 1, dev)
    qd, kd, vd = (t.contiguous().to(BF16).to(dev) for t in (q, k, v))
    mass = torch.zeros(L, 1, dtype=torch.float32, device=dev) if mass_on else None
    if mode == 'plain':
        op = ops.mem_read_attn(qd, kd, vd, out, ws, Lq=L, ldq=C, ldkv=C, ldo=C, nchunks=1, lk_single=Lk)
    else:
        tab = ops.make_chunk_table([(0, 0, Lk, -1, 0)]).to(dev)
        op = ops.mem_read_attn(qd, kd, vd, out, ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=Lk * C, chunks=tab, nchunks=1,
                               lk_single=Lk, mass=mass, T=1)
    ops.run(op)
    torch.cuda.synchronize()
    return out.float().cpu()


g = torch.Generator().manual_seed(3)
for L in (64, 128):
    q = torch.randn(L, C, generator=g).to(BF16).float()
    kr = torch.randn(L, C, generator=g).to(BF16).float()
    vr = torch.randn(L, C, generator=g).to(BF16).float()
    idx = torch.arange(L, dtype=torch.float32)[:, None].expand(L, C).contiguous()
    cases = {'V=1,K rand (expect 1)': (kr, torch.ones(L, C)),
             'K=0,V=key index (expect mean)': (torch.zeros(L, C), idx),
             'K rand,V=key index': (kr, idx),
             'K rand,V rand': (kr, vr)}
    for name, (k, v) in cases.items():
        ref = ref_attn(q, k, v)
        for mode, mass_on in (('plain', False), ('table', False), ('table', True)):
            out = launch(q, k, v, mode, mass_on)
            err = (out - ref).abs()
            print(f'L={L} {name:32s} {mode:5s} mass={int(mass_on)}: max err {err.max():8.4f}  out[0,:4]={[round(x, 3) for x in out[0, :4].tolist()]} '
                  f'ref[0,:4]={[round(x, 3) for x in ref[0, :4].tolist()]}  out[5,32:35]={[round(x, 3) for x in out[5, 32:35].tolist()]} '
                  f'ref={[round(x, 3) for x in ref[5, 32:35].tolist()]}', flush=True)
