"""GPU parity of the DeAOT kernels (gated propagation attention, 15x15 local flavour, SiLU / column-range GEMM epilogues)
against the reference's golden vectors (tests/golden/deaot_ops.npz), the CPU oracle and torch fp32 primitives."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN
from test_hip_ops import assert_close, rb, seeded

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


@pytest.fixture(scope='module')
def w():
    from rmem_ocu_amd.weights import synth_state_dict
    return synth_state_dict(0, model='deaot')


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(GOLDEN, 'deaot_ops.npz'))


def chunk_table(rows, dev):
    from rmem_ocu_amd import ops
    return ops.make_chunk_table(rows).to(dev)


def frame_rows(T, L, splits, pes=None):
    per = (math.ceil(L / splits) + 63) // 64 * 64
    rows = []
    for t in range(T):
        kb = 0
        while kb < L:
            rows.append((t, kb, min(per, L - kb), -1 if pes is None else pes[t], t))
            kb += per
    return rows


def tail(x, wts, p, h, wd, dev):
    """dw_conv + projection of a GatedPropagation module on a bf16 [L, 1024] device tensor (attention.py:210-211)."""
    from rmem_ocu_amd import ops
    L = h * wd
    dw = wts[p + '.dw_conv.conv.weight'].reshape(1024, 25).t().contiguous().to(dev)
    y1 = torch.empty(L, 1024, dtype=BF16, device=dev)
    y2 = torch.empty(L, 512, dtype=F32, device=dev)
    ops.run([ops.dwconv5x5(x, dw, y1, H=h, W=wd, C=1024),
             ops.linear(y1, wts[p + '.projection.weight'].to(BF16).to(dev), wts[p + '.projection.bias'].to(dev), y2, M=L, K=1024, N=512)])
    return y2


@pytest.mark.parametrize('T,splits', [(1, 1), (4, 1), (9, 1), (9, 2)])
def test_gated_attn_golden(dev, w, g, T, splits):
    """The reference's long_term_attn module on seeded inputs: attention x U here, dw_conv + projection by the conv ops."""
    from rmem_ocu_amd import ops
    h, wd = 9, 11
    L = h * wd
    q, k = seeded(4000 + T, (L, 128)), seeded(4100 + T, (T, L, 128))
    v, u = seeded(4200 + T, (T, L, 1024)), seeded(4300 + T, (L, 1024))
    rows = frame_rows(T, L, splits)
    ws = ops.gated_workspace(L, 1024, T, L, len(rows), dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    mass = torch.zeros(L, T, dtype=F32, device=dev)
    ua = u[:, :512].contiguous().to(BF16).to(dev)
    ub = u[:, 512:].contiguous().to(BF16).to(dev)
    ops.run(ops.gated_attn(q.to(BF16).to(dev), k.to(BF16).to(dev), v.to(BF16).to(dev), ua, out, ws, Lq=L, DV=1024, ldq=128, ldk=128,
                           ldv=1024, ldua=512, ldo=1024, k_slot_stride=L * 128, v_slot_stride=L * 1024, chunks=chunk_table(rows, dev),
                           nchunks=len(rows), frames=T, keys_per_frame=L, u_b=ub, ldub=512, usplit=512, mass=mass))
    y = tail(out, w, 'LSTT.layers.0.long_term_attn', h, wd, dev)
    torch.cuda.synchronize()
    assert_close(y, torch.from_numpy(g[f'gp_T{T}_out']), 2e-2, f'gated attn T={T}')
    assert (mass.cpu() - torch.from_numpy(g[f'gp_T{T}_mass'])).abs().max().item() < 4e-3


def ref_gated(q, k, v, u, pe_cur=None, pe_mem=None, slots=None):
    """fp32 torch reference of softmax((q + pe_cur)(k + pe_mem[slot])^T / sqrt(128)) v * u; k [T, L, 128], v [T, L, DV]."""
    T, L, _ = k.shape
    qq = q + (pe_cur if pe_cur is not None else 0)
    kk = k + (pe_mem[slots][:, None, :] if pe_mem is not None else 0)
    s = (qq / math.sqrt(128.0)) @ kk.reshape(T * L, 128).t()
    a = torch.softmax(s, dim=-1)
    return (a @ v.reshape(T * L, -1)) * u, a.view(-1, T, L).sum(2)


@pytest.mark.parametrize('T,L,splits', [(1, 99, 1), (3, 200, 2), (5, 99, 1), (9, 1674, 4)])
def test_gated_attn_temporal_pe(dev, T, L, splits):
    """Temporal embedding on both sides, slot table for T > 4, ragged last tiles, cfg-2 size; vs a torch fp32 reference."""
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.runtime import temporal_slots
    q, k = rb(seeded(10 + T, (L, 128), 1.5)), rb(seeded(20 + T, (T, L, 128), 1.5))
    v, u = rb(seeded(30 + T, (T, L, 1024))), rb(seeded(40 + T, (L, 1024)))
    pe_cur, pe_mem = seeded(50, (128,), 0.3), seeded(51, (4, 128), 0.3)
    slots = temporal_slots(T)
    qd, kd, vd, ud = (t.to(dev) for t in (q, k, v, u))
    # the kernel rounds (q + pe_cur) * scale to bf16 and adds the memory PE as an fp32 logit bias; mirror the first rounding
    ref, mass_ref = ref_gated(qd, kd, vd, ud, pe_cur.to(dev), pe_mem.to(dev), slots)
    rows = frame_rows(T, L, splits, slots)
    ws = ops.gated_workspace(L, 1024, T, L, len(rows), dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    mass = torch.zeros(L, T, dtype=F32, device=dev)
    ops.run(ops.gated_attn(qd.to(BF16), kd.to(BF16), vd.to(BF16), ud.to(BF16), out, ws, Lq=L, DV=1024, ldq=128, ldk=128, ldv=1024,
                           ldua=1024, ldo=1024, k_slot_stride=L * 128, v_slot_stride=L * 1024, chunks=chunk_table(rows, dev),
                           nchunks=len(rows), frames=T, keys_per_frame=L, pe_cur=pe_cur.to(dev), pe_mem=pe_mem.to(dev), mass=mass))
    torch.cuda.synchronize()
    assert_close(out, ref, 2e-2, f'gated attn pe T={T} L={L}')
    assert (mass - mass_ref).abs().max().item() < 4e-3
    assert (mass.sum(1) - 1).abs().max().item() < 1e-4


def test_gated_attn_permuted_slots(dev):
    """Bank slots in a different physical order than the logical frame order, a free slot in between."""
    from rmem_ocu_amd import ops
    T, L, S = 3, 150, 5
    q, k = rb(seeded(61, (L, 128))), rb(seeded(62, (T, L, 128)))
    v, u = rb(seeded(63, (T, L, 1024))), rb(seeded(64, (L, 1024)))
    phys = [4, 0, 2]
    kb = torch.full((S, L, 128), float('nan'))
    vb = torch.full((S, L, 1024), float('nan'))
    for t, s in enumerate(phys):
        kb[s], vb[s] = k[t], v[t]
    rows = [(phys[t], 0, L, -1, t) for t in range(T)]
    ws = ops.gated_workspace(L, 1024, T, L, T, dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    ops.run(ops.gated_attn(q.to(BF16).to(dev), kb.to(BF16).to(dev), vb.to(BF16).to(dev), u.to(BF16).to(dev), out, ws, Lq=L, DV=1024,
                           ldq=128, ldk=128, ldv=1024, ldua=1024, ldo=1024, k_slot_stride=L * 128, v_slot_stride=L * 1024,
                           chunks=chunk_table(rows, dev), nchunks=T, frames=T, keys_per_frame=L))
    torch.cuda.synchronize()
    ref, _ = ref_gated(q, k, v, u)
    assert_close(out, ref, 2e-2, 'permuted slots')


def test_gated_attn_extreme_logits(dev):
    """One key dominates by ~2^60 in the log2 domain, one query row is all zeros: exact-max softmax must stay finite."""
    from rmem_ocu_amd import ops
    L = 130
    q, k = rb(seeded(71, (L, 128))), rb(seeded(72, (1, L, 128)))
    q[5] = 0
    q[7] = k[0, 100] * 30
    v, u = rb(seeded(73, (1, L, 1024))), torch.ones(L, 1024)
    ws = ops.gated_workspace(L, 1024, 1, L, 2, dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    ops.run(ops.gated_attn(q.to(BF16).to(dev), k.to(BF16).to(dev), v.to(BF16).to(dev), u.to(BF16).to(dev), out, ws, Lq=L, DV=1024, ldq=128,
                           ldk=128, ldv=1024, ldua=1024, ldo=1024, nchunks=2, frames=1, keys_per_frame=L))
    torch.cuda.synchronize()
    ref, _ = ref_gated(q, k, v, u)
    assert_close(out, ref, 2e-2, 'extreme logits')
    assert_close(out[7], v[0, 100], 1e-2, 'dominant key row')


@pytest.mark.parametrize('scale', [5, 8, 30])
def test_gated_attn_sampled_reference_and_its_fallback(dev, scale):
    """The softmax reference comes from a sample of the keys (first tile of every table row).  A dominant key OUTSIDE the sample:
    scale 5 leaves it 2^58 above the sampled reference (probabilities far above 1, no fallback); scale 8 (2^93) and scale 30 (2^348:
    the probability itself overflows) are beyond the 2^64 guard, so pass 1 raises the guard word and the exact two-pass redo runs --
    all must give the softmax of the reference implementation.  Three frames, ragged tiles, with the per-frame probability mass."""
    from rmem_ocu_amd import ops
    T, L = 3, 200
    q, k = rb(seeded(81, (L, 128))), rb(seeded(82, (T, L, 128)))
    q[11] = k[1, 150] * scale                   # key 150 of frame 1: third tile of its row
    q[12] = 0
    v, u = rb(seeded(83, (T, L, 1024))), rb(seeded(84, (L, 1024)))
    rows = frame_rows(T, L, 1)
    ws = ops.gated_workspace(L, 1024, T, L, len(rows), dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    mass = torch.zeros(L, T, dtype=F32, device=dev)
    ops.run(ops.gated_attn(q.to(BF16).to(dev), k.to(BF16).to(dev), v.to(BF16).to(dev), u.to(BF16).to(dev), out, ws, Lq=L, DV=1024, ldq=128,
                           ldk=128, ldv=1024, ldua=1024, ldo=1024, k_slot_stride=L * 128, v_slot_stride=L * 1024, chunks=chunk_table(rows, dev),
                           nchunks=len(rows), frames=T, keys_per_frame=L, mass=mass))
    torch.cuda.synchronize()
    ref, mass_ref = ref_gated(q.to(dev), k.to(dev), v.to(dev), u.to(dev))
    assert torch.isfinite(out.float()).all()
    assert_close(out, ref, 2e-2, f'sampled reference, scale {scale}')
    assert_close(out[11], (v[1, 150] * u[11]).to(dev), 2e-2, 'dominant key row')
    assert (mass - mass_ref).abs().max().item() < 4e-3
    flag = ws.view(torch.int32)[ops._lib.lib().rmem_gated_attn_workspace_bytes(L, 1024, T, L, len(rows)) // 4 - 64].item()
    assert flag == (0 if scale == 5 else 1), 'guard word of the sampled reference'


def test_self_gated_attn_golden(dev, w, g):
    """The reference's self_attn module (use_linear True): fused [QK | V | U] GEMM with SiLU from column 128 on, attention
    over one key frame without a chunk table, dw_conv + projection."""
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.pack import pack_deaot_self
    h, wd = 9, 11
    L = h * wd
    x = seeded(4500, (L, 512))
    Wf, bf = pack_deaot_self(w, 'LSTT.layers.0.self_attn')
    qvu = torch.empty(L, 2176, dtype=BF16, device=dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    ws = ops.gated_workspace(L, 1024, 1, L, 8, dev)
    ops.run([ops.linear(x.to(BF16).to(dev), Wf.to(dev), bf.to(dev), qvu, M=L, K=512, N=2176, relu=3, act_begin=128),
             ops.gated_attn(qvu, qvu, qvu.view(-1)[128:], qvu.view(-1)[1152:], out, ws, Lq=L, DV=1024, ldq=2176, ldk=2176, ldv=2176,
                            ldua=2176, ldo=1024, nchunks=8, frames=1, keys_per_frame=L)])
    y = tail(out, w, 'LSTT.layers.0.self_attn', h, wd, dev)
    torch.cuda.synchronize()
    assert_close(y, torch.from_numpy(g['gp_self_out']), 2e-2, 'self gated attn')


@pytest.mark.parametrize('h,wd,key,step', [(9, 11, 'lgp_out', 1), (18, 23, 'lgp_big_out', 2)])
def test_local_gated_attn_golden(dev, w, g, h, wd, key, step):
    """The reference's short_term_attn module: relative embedding GEMM (fp32 out), window attention, dw_conv + projection."""
    from rmem_ocu_amd import ops
    L = h * wd
    base = 4400 if h == 9 else 4410
    q = seeded(base, (1, 128, h, wd))[0].permute(1, 2, 0).reshape(L, 128)
    k = seeded(base + 1, (1, 128, h, wd))[0].permute(1, 2, 0).reshape(L, 128)
    v = seeded(base + 2, (1, 1024, h, wd))[0].permute(1, 2, 0).reshape(L, 1024)
    u = seeded(base + 3, (L, 1024))
    p = 'LSTT.layers.0.short_term_attn'
    qd = q.contiguous().to(BF16).to(dev)
    rel = torch.zeros(L, 256, dtype=F32, device=dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    ws = ops.gated_workspace(L, 1024, 1, L, 8, dev)
    ops.run([ops.linear(qd, w[p + '.relative_emb_k.weight'].reshape(225, 128).to(BF16).to(dev), w[p + '.relative_emb_k.bias'].to(dev), rel,
                        M=L, K=128, N=225, ldo=256),
             ops.local_gated_attn(qd, k.contiguous().to(BF16).to(dev), v.contiguous().to(BF16).to(dev), rel, u.to(BF16).to(dev), out, ws,
                                  H=h, W=wd, DV=1024, ldq=128, ldk=128, ldv=1024, ldrel=256, ldua=1024, ldo=1024)])
    y = tail(out, w, p, h, wd, dev)
    torch.cuda.synchronize()
    assert_close(y[::step], torch.from_numpy(g[key]), 2e-2, f'local gated attn {h}x{wd}')


def test_local_gated_attn_cfg2_vs_oracle(dev, w):
    """31 x 54 tokens (cfg 2): several query tiles, window bands that skip key tiles; attention x U only, vs the oracle."""
    from oracle import deaot_cpu as D
    from rmem_ocu_amd import ops
    h, wd = 31, 54
    L = h * wd
    q2, k2 = rb(seeded(81, (1, 128, h, wd))), rb(seeded(82, (1, 128, h, wd)))
    v2, u = rb(seeded(83, (1, 1024, h, wd))), rb(seeded(84, (L, 1, 1024)))
    p = 'LSTT.layers.1.short_term_attn'
    # oracle up to `agg * u` (attention.py:349): rebuild from its pieces
    rel_ref = F.conv2d(q2, w[p + '.relative_emb_k.weight'], w[p + '.relative_emb_k.bias']).view(1, 1, 225, L)
    mask = 1 - D._pad_unfold(torch.ones(1, 1, h, wd)).view(1, 1, 225, L)
    qk = ((q2 / math.sqrt(128.0)).unsqueeze(2) * D._pad_unfold(k2).view(1, 128, 225, h, wd)).sum(1).view(1, 1, 225, L)
    attn = torch.softmax(qk + rel_ref - mask * 1e8, dim=2)
    ref = (D._pad_unfold(v2).view(1, 1024, 225, L) * attn).sum(2)[0].t() * u[:, 0]
    tok = lambda x: x[0].permute(1, 2, 0).reshape(L, -1).contiguous()   # noqa: E731
    rel = rel_ref[0, 0].t().contiguous()
    rel = F.pad(rel, (0, 31)).contiguous().to(dev)
    out = torch.zeros(L, 1024, dtype=BF16, device=dev)
    ws = ops.gated_workspace(L, 1024, 1, L, 8, dev)
    ops.run(ops.local_gated_attn(tok(q2).to(BF16).to(dev), tok(k2).to(BF16).to(dev), tok(v2).to(BF16).to(dev), rel,
                                 u[:, 0].contiguous().to(BF16).to(dev), out, ws, H=h, W=wd, DV=1024, ldq=128, ldk=128, ldv=1024,
                                 ldrel=256, ldua=1024, ldo=1024))
    torch.cuda.synchronize()
    assert_close(out, ref, 2e-2, 'local gated attn 31x54')


def test_linear_silu_act_begin_ldx(dev):
    """GEMM epilogue: SiLU on columns >= act_begin only; input read with a row stride (a column range of a wider buffer)."""
    from rmem_ocu_amd import ops
    M, K, N = 333, 256, 640
    xw = rb(seeded(91, (M, 512)))
    wt = rb(seeded(92, (N, K), 1 / 16.0))
    b = seeded(93, (N,), 0.1)
    ref = xw[:, 256:] @ wt.t() + b
    ref = torch.cat([ref[:, :128], F.silu(ref[:, 128:])], 1)
    y = torch.zeros(M, N, dtype=BF16, device=dev)
    xd = xw.to(BF16).to(dev)
    ops.run(ops.linear(xd.view(-1)[256:], wt.to(BF16).to(dev), b.to(dev), y, M=M, K=K, N=N, relu=3, act_begin=128, ldx=512))
    torch.cuda.synchronize()
    assert_close(y, ref, 1e-2, 'silu/act_begin/ldx')


@pytest.mark.parametrize('h,wd', [(9, 11), (31, 54)])
def test_gated_attn_fused_dwconv_is_bit_identical(dev, h, wd):
    """dw=...: the combine launch also applies the depth-wise 5x5 (through an LDS tile); same bits as combine + rmem_dwconv5x5."""
    from rmem_ocu_amd import ops
    T, L = 2, h * wd
    q, k = rb(seeded(101, (L, 128))), rb(seeded(102, (T, L, 128)))
    v, u = rb(seeded(103, (T, L, 1024))), rb(seeded(104, (L, 1024)))
    dw = seeded(105, (25, 1024), 0.2).to(dev)
    rows = frame_rows(T, L, 2)
    ws = ops.gated_workspace(L, 1024, T, L, len(rows), dev)
    a, b, c = (torch.zeros(L, 1024, dtype=BF16, device=dev) for _ in range(3))
    common = dict(Lq=L, DV=1024, ldq=128, ldk=128, ldv=1024, ldua=512, ldo=1024, k_slot_stride=L * 128, v_slot_stride=L * 1024,
                  chunks=chunk_table(rows, dev), nchunks=len(rows), frames=T, keys_per_frame=L, u_b=u[:, 512:].contiguous().to(BF16).to(dev),
                  ldub=512, usplit=512)
    args = (q.to(BF16).to(dev), k.to(BF16).to(dev), v.to(BF16).to(dev), u[:, :512].contiguous().to(BF16).to(dev))
    ops.run([ops.gated_attn(*args, a, ws, **common), ops.dwconv5x5(a, dw, b, H=h, W=wd, C=1024)])
    ops.run(ops.gated_attn(*args, c, ws, dw=dw, H=h, W=wd, **common))
    torch.cuda.synchronize()
    assert torch.equal(b, c)
