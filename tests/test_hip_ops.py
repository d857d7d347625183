"""GPU parity of every librmem_hip.so kernel against the CPU oracle / torch fp32 primitives,
on the same seeded inputs (inputs are rounded to bf16 first, so the comparison isolates the
kernel's own arithmetic: bf16 operands, fp32 accumulation, bf16 or fp32 store)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def seeded(seed, shape, scale=1.0):
    rng = np.random.Generator(np.random.PCG64([seed, 0xC0FFEE]))
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * np.float32(scale))


def rb(t):  # round through bf16
    return t.to(BF16).to(F32)


def assert_close(got, ref, rtol, what=''):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), what
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err <= rtol * scale, f'{what}: max err {err:.4g} vs scale {scale:.4g} (rtol {rtol})'


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda', 0)


def nhwc(x):  # [1,C,H,W] fp32 -> [H*W, C] bf16
    return x[0].permute(1, 2, 0).reshape(-1, x.shape[1]).contiguous().to(BF16)


def pack_w(w, cin_pad=0):
    w = w.permute(0, 2, 3, 1)
    if cin_pad > w.shape[-1]:
        w = F.pad(w, (0, cin_pad - w.shape[-1]))
    return w.contiguous().to(BF16)


CONV_CASES = [
    # H, W, Cin, Cout, k, stride, pad, relu, residual, out_f32
    (31, 54, 64, 64, 3, 1, 1, True, None, False),
    (31, 54, 256, 64, 1, 1, 0, True, None, False),
    (31, 54, 64, 256, 1, 1, 0, True, 'bf16', False),
    (61, 107, 128, 128, 3, 2, 1, True, None, False),
    (61, 107, 256, 512, 1, 2, 0, False, None, False),
    (97, 129, 8, 64, 7, 2, 3, True, None, False),
    (97, 129, 16, 256, 17, 16, 8, False, None, False),
    (1674, 1, 256, 256, 1, 1, 0, False, 'f32', True),
    (1674, 1, 1024, 256, 1, 1, 0, False, None, True),
    (1674, 1, 256, 1024, 1, 1, 0, False, None, False),
    (500, 1, 128, 11, 1, 1, 0, False, None, True),
    (121, 213, 128, 128, 3, 1, 1, False, None, False),
    (31, 54, 256, 256, 3, 1, 1, True, None, False),          # layer3 3x3: split-K
    (1674, 1, 1024, 256, 1, 1, 0, True, 'bf16', False),      # layer3 conv3 + residual: split-K
    (481, 849, 16, 256, 17, 16, 8, False, None, False),      # id bank at cfg-2 size: split-K, K = 4624
    (1674, 1, 256, 256, 1, 1, 0, False, 'f32', True),
    (130, 131, 64, 192, 3, 1, 1, True, None, False),         # 128x128 tiles with a ragged last row tile and a half-empty column tile
    (130, 131, 512, 192, 1, 1, 0, True, 'bf16', False),      # K >= 512: producer / consumer 128x128 form (loader waves + MFMA waves), 1x1,
                                                             # ragged last row tile, half-empty column tile, residual epilogue
    (45, 52, 16, 64, 5, 3, 2, False, None, False),           # row-run form with 2 k-steps per filter row (KW * Cin = 80), stride 3
    (37, 41, 24, 64, 3, 2, 1, True, None, False),            # Cin = 24: neither fast form (row run of 72 elements is allowed: spr = 2)
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv2d(dev, case):
    from rmem_ocu_amd import ops
    H, W, Cin, Cout, k, s, p, relu, res, out_f32 = case
    x = rb(seeded(1, (1, Cin, H, W)))
    w = rb(seeded(2, (Cout, Cin, k, k), 1.0 / (Cin * k * k) ** 0.5))
    b = seeded(3, (Cout,), 0.1)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    r = None
    if res:
        r = seeded(4, (1, Cout, Ho, Wo))
        r = rb(r) if res == 'bf16' else r
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    ldo = 16 if Cout == 11 else Cout
    y = torch.zeros(Ho * Wo, ldo, dtype=F32 if out_f32 else BF16, device=dev)
    y2 = torch.zeros(Ho * Wo, Cout, dtype=BF16, device=dev)
    rd = None
    if r is not None:
        rd = r[0].permute(1, 2, 0).reshape(-1, Cout).contiguous().to(dev)
        rd = rd.to(BF16) if res == 'bf16' else rd
    refm = ref[0].permute(1, 2, 0).reshape(-1, Cout)
    pre = F.conv2d(x, w, b, stride=s, padding=p)[0].permute(1, 2, 0).reshape(-1, Cout)
    ws = torch.empty(16 * 1674 * 256, dtype=F32, device=dev)
    for use_ws in (None, ws):          # without / with the split-K workspace (engaged for few-tile, deep-K problems)
        y.zero_(); y2.zero_()
        op = ops.conv2d(nhwc(x).to(dev), pack_w(w).to(dev), b.to(dev), y, H=H, W=W, Cin=Cin, Cout=Cout, KH=k, KW=k, stride=s, pad=p,
                        residual=rd, y2=y2, relu=relu, ldo=ldo, ws=use_ws)
        ops.run(op)
        torch.cuda.synchronize()
        assert_close(y[:, :Cout], refm, 1e-2, f'conv {case} ws={use_ws is not None}')
        assert_close(y2, pre, 1e-2, f'conv y2 {case}')
        if ldo > Cout:
            assert y[:, Cout:].abs().max().item() == 0


@pytest.mark.parametrize('T,L', [(1, 42), (2, 42), (5, 42), (8, 42), (12, 42), (3, 300), (8, 1674)])
def test_mem_read_attn_vs_oracle(dev, T, L, synth_weights):
    """a1 + a2 + a3: long-term attention with temporal PE and the per-frame mass output."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.runtime import temporal_slots
    C = 256
    q, k, v = rb(seeded(10 + T, (L, C))), rb(seeded(20 + T, (T, L, C))), rb(seeded(30 + T, (T, L, C)))
    pe_cur, pe_mem = synth_weights['cur_pos_emb'].view(-1), synth_weights['mem_pos_emb']
    slots = temporal_slots(T)
    # oracle: explicit softmax path of attention.py:45-64 on (Q + cur_pe), (K + pe[slot])
    Qh = ((q + pe_cur) / 32 ** 0.5).view(L, 8, 32).permute(1, 0, 2)
    Kh = (k + pe_mem[slots][:, None, :]).reshape(T * L, 8, 32).permute(1, 2, 0)
    Vh = v.reshape(T * L, 8, 32).permute(1, 0, 2)
    attn = torch.softmax(Qh @ Kh, dim=-1)
    ref = (attn @ Vh).permute(1, 0, 2).reshape(L, C)
    ref_mass = attn.view(8, L, T, L).mean(0).sum(2)
    # HIP: bank with shuffled physical slots
    S = T + 2
    perm = list(np.random.RandomState(T).permutation(S)[:T])
    bank_k = torch.zeros(S, L, C, dtype=BF16, device=dev)
    bank_v = torch.zeros(S, L, C, dtype=BF16, device=dev)
    for t, s in enumerate(perm):
        bank_k[s] = k[t].to(dev)
        bank_v[s] = v[t].to(dev)
    splits = 2 if T <= 4 else 1
    per = (L + splits - 1) // splits
    rows = [(int(perm[t]), j * per, min(per, L - j * per), slots[t], t) for t in range(T) for j in range(splits)]
    chunks = ops.make_chunk_table(rows).to(dev)
    out = torch.zeros(L, C, dtype=BF16, device=dev)
    mass = torch.zeros(L, T, dtype=F32, device=dev)
    ws = ops.attn_workspace(L, 8, len(rows), dev)
    ops.run(ops.mem_read_attn(q.to(BF16).to(dev), bank_k, bank_v, out, ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=L * C,
                              chunks=chunks, nchunks=len(rows), pe_cur=pe_cur.to(dev), pe_mem=pe_mem.to(dev).contiguous(),
                              mass=mass, T=T))
    torch.cuda.synchronize()
    assert_close(out, ref, 2e-2, 'attention out')
    assert_close(mass, ref_mass, 2e-2, 'attention mass')
    assert abs(mass.sum(1).mean().item() - 1.0) < 1e-3


def test_mem_read_full_size_properties(dev, synth_weights):
    """BASELINE configuration of the memory read (HW = 1674 tokens, T = 8 bank frames, 4 clips per launch) through the
    size-independent properties of the operation, no oracle needed: (1) the per-frame probability mass of every query sums
    to 1; (2) the result does not depend on which physical bank slots hold the frames nor on how frames are cut into key
    ranges; (3) the read is linear in V; (4) a second launch is bit-identical; (5) the 4-clip launch equals 4 launches."""
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.runtime import temporal_slots
    L, T, C, B, S = 1674, 8, 256, 4, 9
    slots = temporal_slots(T)
    q = rb(seeded(301, (B, L, C))).to(BF16).to(dev)
    kb = rb(seeded(302, (B * S, L, C))).to(BF16).to(dev)
    vb = rb(seeded(303, (B * S, L, C))).to(BF16).to(dev)
    pe_cur = synth_weights['cur_pos_emb'].view(-1).to(dev)
    pe_mem = synth_weights['mem_pos_emb'].to(dev).contiguous()
    order = [list(np.random.RandomState(c).permutation(S)[:T]) for c in range(B)]      # frame t of clip c lives in slot order[c][t]

    def read(kbank, vbank, order, splits, clips=range(B), grouped=True):
        per = (L + splits - 1) // splits
        out = torch.zeros(B, L, C, dtype=BF16, device=dev)
        mass = torch.zeros(B, L, T, dtype=F32, device=dev)
        rows = {c: [(c * S + int(order[c][t]), j * per, min(per, L - j * per), slots[t], t) for t in range(T) for j in range(splits)]
                for c in clips}
        n = T * splits
        ws = ops.attn_workspace(L, 8, n, dev, nclips=B)
        if grouped:
            tab = ops.make_chunk_table([r for c in clips for r in rows[c]]).to(dev)
            ops.run(ops.mem_read_attn(q, kbank, vbank, out, ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=L * C, chunks=tab, nchunks=n,
                                      pe_cur=pe_cur, pe_mem=pe_mem, mass=mass, T=T, nclips=B, q_cs=L * C, out_cs=L * C))
        else:
            for c in clips:
                tab = ops.make_chunk_table(rows[c]).to(dev)
                ops.run(ops.mem_read_attn(q[c], kbank, vbank, out[c], ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=L * C, chunks=tab,
                                          nchunks=n, pe_cur=pe_cur, pe_mem=pe_mem, mass=mass[c], T=T))
        torch.cuda.synchronize()
        return out, mass

    out, mass = read(kb, vb, order, 1)
    assert out.float().abs().mean().item() > 1e-3
    assert (mass.sum(2) - 1.0).abs().max().item() < 1e-4, 'probability mass over the bank frames must sum to 1'      # (1)
    assert mass.min().item() >= 0.0
    out2, mass2 = read(kb, vb, order, 1)
    assert torch.equal(out, out2) and torch.equal(mass, mass2), 'two launches differ'                                   # (4)
    # (5) the launch heuristics give a single clip more key groups than a clip of four (fp32 summation order differs)
    outc, massc = read(kb, vb, order, 1, grouped=False)
    assert_close(outc, out.float().cpu(), 1e-2, 'clip group vs per-clip launches')
    # (P is rounded to bf16 relative to the group's own softmax reference, so another grouping rounds other bits)
    assert (massc - mass).abs().max().item() < 1e-4
    # ... and the grouping itself must not matter: one workgroup walking all 8 frames (no partials) vs one frame per workgroup
    import os
    res = {}
    for tgt in ('1', '1000000'):
        os.environ['RMEM_ATTN_WGS'] = tgt
        try:
            res[tgt] = read(kb, vb, order, 1)
        finally:
            del os.environ['RMEM_ATTN_WGS']
    assert_close(res['1'][0], res['1000000'][0].float().cpu(), 1e-2, 'all rows in one workgroup vs one row per workgroup')
    assert (res['1'][1] - res['1000000'][1]).abs().max().item() < 1e-4
    assert_close(res['1'][0], out.float().cpu(), 1e-2, 'one group vs default grouping')
    # (2) move every frame to another physical slot, and cut frames into 3 key ranges
    order2 = [list(np.random.RandomState(100 + c).permutation(S)[:T]) for c in range(B)]
    kb2, vb2 = torch.zeros_like(kb), torch.zeros_like(vb)
    for c in range(B):
        for t in range(T):
            kb2[c * S + int(order2[c][t])] = kb[c * S + int(order[c][t])]
            vb2[c * S + int(order2[c][t])] = vb[c * S + int(order[c][t])]
    outp, massp = read(kb2, vb2, order2, 1)
    assert torch.equal(out, outp) and torch.equal(mass, massp), 'result depends on the physical bank slots'
    outs, masss = read(kb, vb, order, 3)
    assert_close(outs, out.float().cpu(), 1e-2, 'frames cut into 3 key ranges')
    assert (masss - mass).abs().max().item() < 1e-4
    # (3) linearity in V (K fixed): read(V1 + V2) = read(V1) + read(V2) up to the bf16 rounding of the three outputs
    v2 = rb(seeded(304, (B * S, L, C))).to(BF16).to(dev)
    vsum = (vb.float() + v2.float())
    o1, _ = read(kb, vb, order, 1)
    o2, _ = read(kb, v2, order, 1)
    o12, _ = read(kb, vsum.to(BF16), order, 1)
    # the bf16 rounding of V1 + V2 is itself a perturbation of 2^-9 relative: compare against the read of the ROUNDED sum
    lin = o1.float() + o2.float()
    assert_close(o12, lin.cpu(), 2e-2, 'linearity in V')


def test_plain_attn_split_keys(dev):
    """a4 / a5 core: one key frame split into 4 key ranges, no temporal PE, strided q/k/v views."""
    from rmem_ocu_amd import ops
    L, C = 1674, 256
    qkv = rb(seeded(40, (L, 3 * C)))
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    Qh = (q / 32 ** 0.5).reshape(L, 8, 32).permute(1, 0, 2)
    ref = (torch.softmax(Qh @ k.reshape(L, 8, 32).permute(1, 2, 0), -1) @ v.reshape(L, 8, 32).permute(1, 0, 2)).permute(1, 0, 2).reshape(L, C)
    d = qkv.to(BF16).to(dev)
    out = torch.zeros(L, C, dtype=BF16, device=dev)
    ws = ops.attn_workspace(L, 8, 4, dev)
    ops.run(ops.mem_read_attn(d, d.view(-1)[C:], d.view(-1)[2 * C:], out, ws, Lq=L, ldq=3 * C, ldkv=3 * C, ldo=C, nchunks=4, lk_single=L))
    torch.cuda.synchronize()
    assert_close(out, ref, 2e-2, 'plain attention')


def test_attention_forced_rescale(dev):
    """Online-softmax rescale branch: a late key tile whose logits jump far above everything before it."""
    from rmem_ocu_amd import ops
    L, C = 200, 256
    q, k, v = rb(seeded(50, (L, C))), rb(seeded(51, (L, C))), rb(seeded(52, (L, C)))
    k[150:] = rb(q[:50] * 4.0)          # keys 150.. align with queries 0..49: large logits late in the stream
    Qh = (q / 32 ** 0.5).reshape(L, 8, 32).permute(1, 0, 2)
    ref = (torch.softmax(Qh @ k.reshape(L, 8, 32).permute(1, 2, 0), -1) @ v.reshape(L, 8, 32).permute(1, 0, 2)).permute(1, 0, 2).reshape(L, C)
    out = torch.zeros(L, C, dtype=BF16, device=dev)
    ws = ops.attn_workspace(L, 8, 1, dev)
    ops.run(ops.mem_read_attn(q.to(BF16).to(dev), k.to(BF16).to(dev), v.to(BF16).to(dev), out, ws, Lq=L, ldq=C, ldkv=C, ldo=C,
                              nchunks=1, lk_single=L))
    torch.cuda.synchronize()
    assert_close(out, ref, 2e-2, 'rescale branch')


@pytest.mark.parametrize('wgs', ['1', '1000000'])
def test_attention_late_dominant_key_takes_safe_pass(dev, synth_weights, wgs):
    """The fast pass fixes the softmax reference at the first tile's maximum.  A key far down the stream whose logit is
    ~2^100 above it overflows the fast pass's sums; the workgroup must notice and redo its rows with the online-softmax pass.
    Memory-read flavour with 3 frames (rows), mass output, both groupings (all rows in one workgroup / one row each)."""
    import os
    from rmem_ocu_amd import ops
    T, L, C = 3, 200, 256
    g = torch.Generator().manual_seed(77)
    u = torch.nn.functional.normalize(torch.randn(8, 32, generator=g), dim=1).reshape(C)       # one direction per head
    q = rb(seeded(58, (L, C)) * 0.5 + 20.0 * u)
    k = rb(seeded(59, (T, L, C)) * 0.5)
    v = rb(seeded(60, (T, L, C)))
    k[2, 150] = rb(20.0 * u)                     # frame 2, key 150: q.k = 400 per head -> 400 / sqrt(32) * log2(e) = 102 in log2 units
    Qh = (q / 32 ** 0.5).view(L, 8, 32).permute(1, 0, 2)
    Kh = k.reshape(T * L, 8, 32).permute(1, 2, 0)
    Vh = v.reshape(T * L, 8, 32).permute(1, 0, 2)
    attn = torch.softmax(Qh @ Kh, dim=-1)
    ref = (attn @ Vh).permute(1, 0, 2).reshape(L, C)
    ref_mass = attn.view(8, L, T, L).mean(0).sum(2)
    assert ref_mass[:, 2].min().item() > 0.999            # the spike owns every query
    rows = [(t, 0, L, -1, t) for t in range(T)]
    chunks = ops.make_chunk_table(rows).to(dev)
    out = torch.zeros(L, C, dtype=BF16, device=dev)
    mass = torch.zeros(L, T, dtype=F32, device=dev)
    ws = ops.attn_workspace(L, 8, T, dev)
    os.environ['RMEM_ATTN_WGS'] = wgs
    try:
        ops.run(ops.mem_read_attn(q.to(BF16).to(dev), k.to(BF16).to(dev), v.to(BF16).to(dev), out, ws, Lq=L, ldq=C, ldkv=C, ldo=C,
                                  slot_stride=L * C, chunks=chunks, nchunks=T, mass=mass, T=T))
        torch.cuda.synchronize()
    finally:
        del os.environ['RMEM_ATTN_WGS']
    assert torch.isfinite(out.float()).all() and torch.isfinite(mass).all()
    assert_close(out, ref, 2e-2, 'late dominant key')
    assert_close(mass, ref_mass, 2e-2, 'late dominant key: mass')


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
@pytest.mark.parametrize('wgs', ['1', '1000000'])
def test_attention_all_logits_far_below_zero_with_ragged_rows(dev, wgs, dtype):
    """Every real key scores far BELOW zero (q . k ~ -150 in log2 units: queries and keys point in opposite directions, as the
    logits of a trained model may), and every row ends in a ragged tile (200 keys = 3 x 64 + 8) plus one EMPTY row: the zero
    keys the DMA pads a tile with score q . 0 = 0, i.e. 2^150 above everything real.  They must have no part in the maximum,
    the row sums, the mass or O (the clip of tests/test_hip_engine.py::test_n2_bank_clip is such a case)."""
    import os
    from rmem_ocu_amd import ops
    e16, rnd = (BF16, rb) if dtype == 'bf16' else (F16, rh)
    T, L, C = 3, 200, 256
    g = torch.Generator().manual_seed(79)
    u = torch.nn.functional.normalize(torch.randn(8, 32, generator=g), dim=1).reshape(C)
    q = rnd(seeded(61, (L, C)) * 0.5 + 24.0 * u)
    k = rnd(seeded(62, (T, L, C)) * 0.5 - 24.0 * u)
    v = rnd(seeded(63, (T, L, C)))
    Qh = (q / 32 ** 0.5).view(L, 8, 32).permute(1, 0, 2)
    logits = Qh @ k.reshape(T * L, 8, 32).permute(1, 2, 0)
    assert logits.max().item() < -60.0                    # natural-log units: < -86 in log2
    attn = torch.softmax(logits, dim=-1)
    ref = (attn @ v.reshape(T * L, 8, 32).permute(1, 0, 2)).permute(1, 0, 2).reshape(L, C)
    ref_mass = torch.cat((attn.view(8, L, T, L).mean(0).sum(2), torch.zeros(L, 1)), 1)
    rows = [(t, 0, L, -1, t) for t in range(T)] + [(0, 0, 0, -1, T)]      # + an empty row (frame T has no keys)
    chunks = ops.make_chunk_table(rows).to(dev)
    out = torch.zeros(L, C, dtype=e16, device=dev)
    mass = torch.zeros(L, T + 1, dtype=F32, device=dev)
    ws = ops.attn_workspace(L, 8, T + 1, dev)
    os.environ['RMEM_ATTN_WGS'] = wgs
    try:
        ops.run(ops.mem_read_attn(q.to(e16).to(dev), k.to(e16).to(dev), v.to(e16).to(dev), out, ws, Lq=L, ldq=C, ldkv=C, ldo=C,
                                  slot_stride=L * C, chunks=chunks, nchunks=T + 1, mass=mass, T=T + 1))
        torch.cuda.synchronize()
    finally:
        del os.environ['RMEM_ATTN_WGS']
    assert torch.isfinite(out.float()).all() and torch.isfinite(mass).all()
    assert_close(out, ref, 2e-2, 'negative logits, ragged rows')
    assert_close(mass, ref_mass, 2e-2, 'negative logits, ragged rows: mass')
    assert mass[:, T].abs().max().item() == 0.0


def test_attention_slow_ramp_no_rescale(dev):
    """Lazy-max path: logits that climb a little every tile (below the rescale threshold per step, far above it
    in total), so P is exponentiated against a stale reference for many tiles."""
    from rmem_ocu_amd import ops
    L, C = 640, 256
    q = rb(seeded(55, (L, C)))
    k = rb(seeded(56, (L, C)) * 0.3 + q.mean(0, keepdim=True) * torch.linspace(0, 6, L)[:, None])
    v = rb(seeded(57, (L, C)))
    Qh = (q / 32 ** 0.5).reshape(L, 8, 32).permute(1, 0, 2)
    ref = (torch.softmax(Qh @ k.reshape(L, 8, 32).permute(1, 2, 0), -1) @ v.reshape(L, 8, 32).permute(1, 0, 2)).permute(1, 0, 2).reshape(L, C)
    for nchunks in (1, 3):
        out = torch.zeros(L, C, dtype=BF16, device=dev)
        ws = ops.attn_workspace(L, 8, nchunks, dev)
        ops.run(ops.mem_read_attn(q.to(BF16).to(dev), k.to(BF16).to(dev), v.to(BF16).to(dev), out, ws, Lq=L, ldq=C, ldkv=C, ldo=C,
                                  nchunks=nchunks, lk_single=L))
        torch.cuda.synchronize()
        assert_close(out, ref, 2e-2, f'ramp nchunks={nchunks}')


def test_layernorm(dev):
    from rmem_ocu_amd import ops
    M = 1674
    a, b = seeded(60, (M, 256)), rb(seeded(61, (M, 256)))
    g, be, pos = 1 + 0.1 * seeded(62, (256,)), 0.1 * seeded(63, (256,)), seeded(64, (M, 256))
    ref = F.layer_norm(a + b, (256,), g, be, 1e-5)
    y = torch.zeros(M, 1024, dtype=BF16, device=dev)
    yp = torch.zeros(M, 256, dtype=BF16, device=dev)
    yf = torch.zeros(M, 256, dtype=F32, device=dev)
    ops.run(ops.layernorm256(a.to(dev), g.to(dev), be.to(dev), M=M, b=b.to(BF16).to(dev), y=y.view(-1)[256:], ldy=1024,
                             pos=pos.to(dev), ypos=yp, yf=yf))
    torch.cuda.synchronize()
    assert_close(yf, ref, 1e-5, 'ln f32')
    assert_close(y[:, 256:512], ref, 1e-2, 'ln bf16 strided')
    assert_close(yp, ref + pos, 1e-2, 'ln + pos')
    assert y[:, :256].abs().max().item() == 0


@pytest.mark.parametrize('M,C,groups,act', [(1674, 1024, 32, 2), (1674, 256, 8, 1), (6527, 128, 8, 1), (300, 128, 8, 0)])
def test_groupnorm(dev, M, C, groups, act):
    from rmem_ocu_amd import ops
    x = rb(seeded(70, (M, C)) * 2 + 0.5)
    g, b = 1 + 0.1 * seeded(71, (C,)), 0.1 * seeded(72, (C,))
    ref = F.group_norm(x.t().reshape(1, C, M, 1), groups, g, b, 1e-5)
    ref = F.relu(ref) if act == 1 else F.gelu(ref) if act == 2 else ref
    y = torch.zeros(M, C, dtype=BF16, device=dev)
    ws = ops.groupnorm_workspace(groups, dev)
    ops.run(ops.groupnorm(x.to(BF16).to(dev), g.to(dev), b.to(dev), y, ws, M=M, C=C, groups=groups, act=act))
    torch.cuda.synchronize()
    assert_close(y, ref.reshape(C, M).t(), 1e-2, 'groupnorm')


def test_dwconv_and_ffn_activation(dev, synth_weights):
    """basic.py:27-35 GN(32)+GELU+DW5x5 against the oracle's gn_gelu_dwconv."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd import ops
    h, w, C = 31, 54, 1024
    x = rb(seeded(80, (h * w, 1, C)))
    p = 'LSTT.layers.0.activation'
    ref = O.gn_gelu_dwconv(x, (h, w), synth_weights, p)[:, 0]
    xd = x[:, 0].to(BF16).to(dev)
    y1 = torch.zeros(h * w, C, dtype=BF16, device=dev)
    y2 = torch.zeros(h * w, C, dtype=BF16, device=dev)
    ws = ops.groupnorm_workspace(32, dev)
    wt = synth_weights[p + '.conv.weight'].view(C, 25).t().contiguous().to(dev)
    ops.run([ops.groupnorm(xd, synth_weights[p + '.gn.weight'].to(dev), synth_weights[p + '.gn.bias'].to(dev), y1, ws, M=h * w, C=C, groups=32, act=2),
             ops.dwconv5x5(y1, wt, y2, H=h, W=w, C=C)])
    torch.cuda.synchronize()
    assert_close(y2, ref, 2e-2, 'gn+gelu+dwconv')


def test_resample_kernels(dev):
    from rmem_ocu_amd import ops
    # image -> NHWC8
    img = seeded(90, (3, 97, 129))
    out = torch.zeros(97 * 129, 8, dtype=BF16, device=dev)
    ops.run(ops.image_to_nhwc8(img.to(dev), out, H=97, W=129))
    assert_close(out[:, :3], rb(img).permute(1, 2, 0).reshape(-1, 3), 1e-6, 'nhwc8')
    assert out[:, 3:].abs().max().item() == 0
    # maxpool
    x = rb(seeded(91, (1, 64, 49, 65)))
    y = torch.zeros(25 * 33, 64, dtype=BF16, device=dev)
    ops.run(ops.maxpool3x3s2(nhwc(x).to(dev), y, H=49, W=65, C=64))
    assert_close(y, nhwc(F.max_pool2d(x, 3, 2, 1)), 1e-6, 'maxpool')
    # bilinear, both align modes
    x = rb(seeded(92, (1, 128, 31, 54)))
    for ac, (ho, wo) in ((True, (61, 107)), (False, (62, 108))):
        y = torch.zeros(ho * wo, 128, dtype=BF16, device=dev)
        ops.run(ops.bilinear(nhwc(x).to(dev), y, Hi=31, Wi=54, Ho=ho, Wo=wo, C=128, align_corners=ac))
        assert_close(y, nhwc(F.interpolate(x, size=(ho, wo), mode='bilinear', align_corners=ac)), 1e-2, f'bilinear ac={ac}')
    # logits post
    lg = seeded(93, (1, 11, 25, 33))
    lgn = torch.zeros(25 * 33, 16)
    lgn[:, :11] = lg[0].permute(1, 2, 0).reshape(-1, 11)
    o = torch.zeros(11, 96, 128, dtype=F32, device=dev)
    lab = torch.zeros(96, 128, dtype=torch.uint8, device=dev)
    ops.run(ops.logits_post(lgn.to(dev), ldl=16, nc=11, keep=3, Hi=25, Wi=33, Ho=96, Wo=128, out=o, label_u8=lab))
    ref = lg.clone()
    ref[:, 4:] = -1e10
    ref = F.interpolate(ref, size=(96, 128), mode='bilinear', align_corners=True)[0]
    assert_close(o[:4], ref[:4], 1e-5, 'logits')
    assert (o[4:] < -1e9).all()
    assert (lab.cpu().long() == ref.argmax(0)).float().mean().item() > 0.9999
    torch.cuda.synchronize()


@pytest.mark.parametrize('geom', [(121, 213, 480, 854, True), (41, 49, 160, 192, True), (120, 214, 480, 856, False),
                                  (25, 33, 40, 50, True)])
def test_logits_label_only_routes_vs_fp32(dev, geom):
    """The LABEL-ONLY route of rmem_logits_post_images -- what every group step of the throughput path runs (k_logits_labels_tile
    for >= 2x upsampling, k_logits_labels4 below that) -- against F.interpolate + argmax in fp32 (engines/aot_engine.py:450-463,
    managers/evaluator.py:430-441): 4 images per launch, ids above `keep` masked, bench geometry 121x213 -> 480x854 included.
    Labels must be equal except where the two best interpolated logits are within fp32 rounding of each other (the kernel blends
    rows then columns with its own operation order)."""
    from rmem_ocu_amd import ops
    Hi, Wi, Ho, Wo, ac = geom
    B, nc, keep = 4, 11, 6
    lg = seeded(97 + Hi, (B, nc, Hi, Wi)) * 3.0
    lgn = torch.zeros(B, Hi * Wi, 16)
    lgn[:, :, :nc] = lg.permute(0, 2, 3, 1).reshape(B, -1, nc)
    lab = torch.full((B, Ho, Wo), 255, dtype=torch.uint8, device=dev)
    labf = torch.full((B, Ho, Wo), -1.0, dtype=F32, device=dev)
    ops.run(ops.logits_post(lgn.to(dev), ldl=16, nc=nc, keep=keep, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, align_corners=ac, label_u8=lab,
                            label_f32=labf, images=B))
    torch.cuda.synchronize()
    ref = lg.clone()
    ref[:, keep + 1:] = -1e10
    up = F.interpolate(ref, size=(Ho, Wo), mode='bilinear', align_corners=ac)          # fp32, as the reference runs it
    top2 = up.topk(2, dim=1).values
    # "tie": the two best blended logits are closer than what fp32 rounding of the source coordinate (up to 213 * 2^-24 in the
    # blend weight) and of the blend itself can move them
    tie = (top2[:, 0] - top2[:, 1]) < 1e-4 * up[:, :keep + 1].abs().amax(1).clamp_min(1.0)
    got = lab.cpu().long()
    assert (got <= keep).all()
    assert torch.equal(got.float(), labf.cpu())
    bad = (got != up.argmax(1)) & ~tie
    assert not bad.any(), f'{int(bad.sum())} labels differ away from ties ({int(tie.sum())} near-ties among {tie.numel()})'
    # one image at a time gives the same labels (clip c of a group = the per-clip engine's call)
    one = torch.zeros(Ho, Wo, dtype=torch.uint8, device=dev)
    for c in (0, B - 1):
        ops.run(ops.logits_post(lgn[c].to(dev), ldl=16, nc=nc, keep=keep, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, align_corners=ac, label_u8=one))
        assert torch.equal(one, lab[c])
    torch.cuda.synchronize()


def test_label_onehot_and_id_bank(dev, synth_weights):
    """a11: label -> nearest resize -> one-hot -> 17x17 s16 conv against the oracle's assign_identity."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.pack import _conv_w
    lab = torch.zeros(96, 128)
    lab[10:50, 20:70] = 1
    lab[40:90, 60:120] = 3
    lab[0:8, 0:8] = 255
    H, W = 97, 129
    near = F.interpolate(lab[None, None], size=(H, W), mode='nearest')
    oh, ign = O.one_hot_mask(near)
    ref = O.assign_identity(oh, ign, synth_weights)[:, 0]
    onehot = torch.zeros(H * W, 16, dtype=BF16, device=dev)
    emb = torch.zeros(7 * 9, 256, dtype=BF16, device=dev)
    w = _conv_w(synth_weights['patch_wise_id_bank.weight'], 16).to(dev)
    ops.run([ops.label_to_onehot16(lab.to(torch.uint8).to(dev), onehot, Hs=96, Ws=128, Hd=H, Wd=W),
             ops.conv2d(onehot, w, synth_weights['patch_wise_id_bank.bias'].to(dev), emb, H=H, W=W, Cin=16, Cout=256, KH=17, KW=17, stride=16, pad=8)])
    torch.cuda.synchronize()
    assert_close(emb, ref, 1.5e-2, 'id emb')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('images,hs,ws,H,W,K,stride,pad', [(1, 96, 128, 97, 129, 17, 16, 8), (3, 480, 854, 481, 849, 17, 16, 8), (2, 64, 96, 64, 96, 16, 16, 0),
                                                          (8, 120, 160, 121, 161, 17, 16, 8)])
def test_label_id_embed_without_onehot_tensor(dev, images, hs, ws, H, W, K, stride, pad, dt):
    """rmem_label_id_embed: the id bank's conv with its one-hot operand built in registers from the label bytes (models/aot.py:139-147,
    engines/aot_engine.py:208-232) against the two launches it replaces (rmem_label_to_onehot16 + rmem_conv2d_nhwc: the same
    products in another summation order: equal up to the output rounding) and against fp32 torch; uint8 and fp32 labels, the ignore
    label 255, labels beyond the class count, the Swin geometry (16 x 16, stride 16, no padding), several images."""
    from rmem_ocu_amd import ops
    g = torch.Generator().manual_seed(123)
    lab = torch.randint(0, 11, (images, hs, ws), generator=g)
    lab[:, : hs // 3, : ws // 2] = torch.randint(0, 4, (images, hs // 3, ws // 2), generator=g)
    lab[:, 0:5, 0:7] = 255
    lab[:, -3:, -9:] = 13                      # beyond the class count: contributes nothing
    w = (seeded(130, (256, 11, K, K), 1.0 / K)).to(dt).float()
    bias = seeded(131, (256,), 0.1)
    near = F.interpolate(lab[:, None].float(), size=(H, W), mode='nearest')[:, 0].long()
    onehot = torch.zeros(images, 11, H, W)
    for c in range(11):
        onehot[:, c] = (near == c).float()
    ref = F.conv2d(onehot, w, bias, stride=stride, padding=pad)
    Ho, Wo = ref.shape[2], ref.shape[3]
    ref = ref.permute(0, 2, 3, 1).reshape(images * Ho * Wo, 256)
    wd = F.pad(w.permute(0, 2, 3, 1), (0, 5)).contiguous().to(dt).to(dev)
    for labels in (lab.to(torch.uint8).to(dev), lab.float().to(dev)):
        scratch = ops.label_id_embed_scratch(images, H, W, pad, dev)
        out = torch.zeros(images * Ho * Wo, 256, dtype=dt, device=dev)
        ops.run(ops.label_id_embed(labels, wd, bias.to(dev), scratch, out, Hs=hs, Ws=ws, H=H, W=W, K=K, stride=stride, pad=pad, images=images))
        oh16 = torch.zeros(images * H * W, 16, dtype=dt, device=dev)
        out2 = torch.zeros_like(out)
        ops.run([ops.label_to_onehot16(labels, oh16, Hs=hs, Ws=ws, Hd=H, Wd=W, images=images),
                 ops.conv2d(oh16, wd, bias.to(dev), out2, H=H, W=W, Cin=16, Cout=256, KH=K, KW=K, stride=stride, pad=pad, batch=images)])
        torch.cuda.synchronize()
        assert torch.equal(scratch[:, pad:pad + H, pad:pad + W].cpu().long(), near.clamp(max=255)) and int(scratch[:, -1].min()) == 255
        assert_close(out, ref, 1e-2, 'id embedding')
        d = (out.float() - out2.float()).abs()
        assert d.max().item() <= 2.0 ** (-7 if dt == torch.bfloat16 else -10) * max(1.0, out2.float().abs().max().item()), d.max().item()


def test_evict_scores(dev):
    from rmem_ocu_amd import ops
    lg = seeded(95, (1, 11, 25, 33))
    lgn = torch.zeros(25 * 33, 16)
    lgn[:, :11] = lg[0].permute(1, 2, 0).reshape(-1, 11)
    mass = seeded(96, (7 * 9, 5)).abs()
    fg = 1 - torch.softmax(F.interpolate(lg, size=(7, 9), mode='bilinear', align_corners=True), 1)[0, 0].flatten()
    ref = (mass * fg[:, None]).sum(0)
    sc = torch.zeros(32 + 64 * 32, dtype=F32, device=dev)
    ops.run(ops.evict_scores(lgn.to(dev), mass.to(dev), sc, ldl=16, nc=11, keep=10, Hi=25, Wi=33, He=7, We=9, T=5))
    torch.cuda.synchronize()
    assert_close(sc[:5], ref, 1e-4, 'evict scores')


def test_layernorm_generic_and_patch_merge(dev, golden_ops):
    from oracle import ref_cpu as O
    from rmem_ocu_amd import ops
    for C in (128, 256, 512, 1024):
        a = seeded(300 + C, (333, C)) * 2 + 0.3
        g, b = 1 + 0.1 * seeded(301, (C,)), 0.1 * seeded(302, (C,))
        y = torch.zeros(333, C, dtype=BF16, device=dev)
        yf = torch.zeros(333, C, dtype=F32, device=dev)
        ops.run(ops.layernorm(a.to(dev), g.to(dev), b.to(dev), M=333, C=C, y=y, yf=yf))
        torch.cuda.synchronize()
        ref = F.layer_norm(a, (C,), g, b, 1e-5)
        assert_close(yf, ref, 2e-5, f'ln{C} f32')
        assert_close(y, ref, 1e-2, f'ln{C} bf16')
    for C, (H, W) in ((128, (9, 11)), (256, (6, 8))):
        x = seeded(310 + C, (1, H * W, C))
        w = {'m.norm.weight': 1 + 0.1 * seeded(311, (4 * C,)), 'm.norm.bias': 0.1 * seeded(312, (4 * C,)),
             'm.reduction.weight': torch.eye(2 * C, 4 * C)}
        xp = x.view(1, H, W, C)
        xp = F.pad(xp, (0, 0, 0, W % 2, 0, H % 2))
        cat = torch.cat([xp[:, 0::2, 0::2], xp[:, 1::2, 0::2], xp[:, 0::2, 1::2], xp[:, 1::2, 1::2]], -1).view(-1, 4 * C)
        ref = F.layer_norm(cat, (4 * C,), w['m.norm.weight'], w['m.norm.bias'], 1e-5)
        y = torch.zeros(ref.shape[0], 4 * C, dtype=BF16, device=dev)
        ops.run(ops.patch_merge_ln(x[0].to(dev), w['m.norm.weight'].to(dev), w['m.norm.bias'].to(dev), y, H=H, W=W, C=C))
        torch.cuda.synchronize()
        assert_close(y, ref, 1e-2, f'patch merge {C}')


@pytest.mark.parametrize('H,W,heads,shift', [(24, 32, 4, 0), (24, 32, 4, 3), (12, 16, 8, 3), (6, 8, 16, 3), (7, 7, 4, 3), (45, 80, 16, 0)])
def test_window_attention_vs_oracle(dev, H, W, heads, shift):
    """a16: (shifted-)window attention incl. padding, roll, mask and relative-position bias against the oracle's
    swin_block pieces (window partition / softmax / reverse on bf16-rounded qkv)."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.pack import LOG2E, swin_window_masks
    from rmem_ocu_amd.weights import swin_relative_position_index
    C, ws = heads * 32, 7
    L = H * W
    qkv = rb(seeded(400 + H, (L, 3 * C)))
    qkv_bias = seeded(401, (3 * C,), 0.3)
    table = seeded(402, (169, heads), 0.5)
    idx = swin_relative_position_index(7)
    # oracle: qkv of padded tokens = bias; roll; partition; attention; reverse; unroll; crop
    pr, pb = (ws - W % ws) % ws, (ws - H % ws) % ws
    g = qkv.view(1, H, W, 3 * C)
    Hp, Wp = H + pb, W + pr
    full = rb(qkv_bias).view(1, 1, 1, -1).expand(1, Hp, Wp, 3 * C).clone()
    full[:, :H, :W] = g
    if shift:
        full = torch.roll(full, shifts=(-shift, -shift), dims=(1, 2))
    xw = O._win_part(full, ws)
    B_, N = xw.shape[0], 49
    q, k, v = xw.view(B_, N, 3, heads, 32).permute(2, 0, 3, 1, 4)
    attn = (q * 32 ** -0.5) @ k.transpose(-2, -1) + table[idx.view(-1)].view(N, N, heads).permute(2, 0, 1).unsqueeze(0)
    if shift:
        mask = O.swin_shift_mask(H, W)
        attn = (attn.view(1, B_, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    o = (torch.softmax(attn, -1) @ v).transpose(1, 2).reshape(B_, N, C)
    y = O._win_rev(o, ws, Hp, Wp)
    if shift:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    ref = y[:, :H, :W].reshape(L, C)
    # HIP
    bias = table[idx.view(-1)].view(49, 49, heads).permute(2, 0, 1)
    tbl = ((bias[None] + swin_window_masks()[:, None]) * LOG2E).contiguous().to(dev)
    out = torch.zeros(L, C, dtype=BF16, device=dev)
    ops.run(ops.window_attn(qkv.to(BF16).to(dev), rb(qkv_bias).to(dev), tbl, out, H=H, W=W, C=C, heads=heads, shift=shift))
    torch.cuda.synchronize()
    assert_close(out, ref, 2e-2, f'window attention {H}x{W} shift {shift}')


@pytest.mark.parametrize('src,dst', [((480, 854), (481, 849)), ((1080, 1920), (577, 1041)), ((120, 160), (120, 160)), ((50, 70), (97, 129))])
def test_frame_ingest(dev, src, dst):
    """f2: uint8 RGB -> bicubic resize -> normalise, fp32 CHW and NHWC8 bf16, against the oracle's restatement of the
    reference data path (cv2 is absent in this image: the oracle side is unpinned for this row)."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd import ops
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, size=(src[0] // 8 + 2, src[1] // 8 + 2, 3)).astype(np.float32)
    img = np.clip(F.interpolate(torch.from_numpy(base).permute(2, 0, 1)[None], size=src, mode='bilinear')[0].permute(1, 2, 0).numpy()
                  + rng.normal(0, 8, size=(*src, 3)), 0, 255).astype(np.uint8)
    ref = torch.from_numpy(O.ingest_rgb8(img, *dst))
    chw = torch.zeros(3, *dst, dtype=F32, device=dev)
    n8 = torch.zeros(dst[0] * dst[1], 8, dtype=BF16, device=dev)
    ops.run(ops.ingest_rgb8(torch.from_numpy(img).to(dev), Hs=src[0], Ws=src[1], Hd=dst[0], Wd=dst[1], out_chw=chw, out_nhwc8=n8))
    torch.cuda.synchronize()
    assert_close(chw, ref, 2e-5, f'ingest {src}->{dst}')
    assert_close(n8[:, :3], rb(ref).permute(1, 2, 0).reshape(-1, 3), 1e-2, 'ingest nhwc8')
    assert n8[:, 3:].abs().max().item() == 0


def test_conv2d_batch_of_images(dev):
    """batch > 1: rows are [image][ho][wo]; every image must equal the single-image launch (3x3 stride 2 with padding, so a
    row index that ignored the image would bleed across image borders)."""
    from rmem_ocu_amd import ops
    B, H, W, Cin, Cout = 3, 23, 31, 64, 128
    x = rb(seeded(11, (B, Cin, H, W)))
    w = rb(seeded(12, (Cout, Cin, 3, 3), 1.0 / (Cin * 9) ** 0.5))
    b = seeded(13, (Cout,), 0.1)
    ref = F.relu(F.conv2d(x, w, b, stride=2, padding=1))
    Ho, Wo = ref.shape[2], ref.shape[3]
    xd = x.permute(0, 2, 3, 1).reshape(B, H * W, Cin).contiguous().to(BF16).to(dev)
    y = torch.zeros(B, Ho * Wo, Cout, dtype=BF16, device=dev)
    ops.run(ops.conv2d(xd, pack_w(w).to(dev), b.to(dev), y, H=H, W=W, Cin=Cin, Cout=Cout, KH=3, KW=3, stride=2, pad=1, relu=True, batch=B))
    y1 = torch.zeros(Ho * Wo, Cout, dtype=BF16, device=dev)
    for i in range(B):
        ops.run(ops.conv2d(xd[i], pack_w(w).to(dev), b.to(dev), y1, H=H, W=W, Cin=Cin, Cout=Cout, KH=3, KW=3, stride=2, pad=1, relu=True))
        torch.cuda.synchronize()
        assert torch.equal(y[i], y1), f'image {i} differs from the single-image launch'
    assert_close(y, ref.permute(0, 2, 3, 1).reshape(B, Ho * Wo, Cout), 1e-2, 'batched conv')


@pytest.mark.parametrize('B,H2,W2,K1,K2,Cout,stride', [(2, 23, 31, 64, 128, 192, 2), (1, 9, 14, 64, 64, 256, 1), (3, 61, 43, 256, 512, 1024, 2)])
def test_conv1x1_dual_bottleneck_tail(dev, B, H2, W2, K1, K2, Cout, stride):
    """relu(conv3(h) + b3 + downsample(x) + bd) of a ResNet bottleneck (encoders/resnet.py:48-68) as one GEMM over [h | x sampled];
    fp32 reference on the bf16-rounded operands.  The 3rd case takes the 128x128 tile, the others the 64x64 one."""
    from rmem_ocu_amd import ops
    Ho, Wo = (H2 - 1) // stride + 1, (W2 - 1) // stride + 1
    h = rb(seeded(51, (B, K1, Ho, Wo)))
    x = rb(seeded(52, (B, K2, H2, W2)))
    w3 = rb(seeded(53, (Cout, K1, 1, 1), 1.0 / K1 ** 0.5))
    wd = rb(seeded(54, (Cout, K2, 1, 1), 1.0 / K2 ** 0.5))
    b3, bd = seeded(55, (Cout,), 0.1), seeded(56, (Cout,), 0.1)
    ref = F.relu(F.conv2d(h, w3, b3) + F.conv2d(x, wd, bd, stride=stride))
    hd = h.permute(0, 2, 3, 1).reshape(B * Ho * Wo, K1).contiguous().to(BF16).to(dev)
    xd = x.permute(0, 2, 3, 1).contiguous().to(BF16).to(dev)
    wcat = torch.cat([w3.reshape(Cout, K1), wd.reshape(Cout, K2)], 1).contiguous().to(BF16).to(dev)
    y = torch.zeros(B * Ho * Wo, Cout, dtype=BF16, device=dev)
    ops.run(ops.conv1x1_dual(hd, xd, wcat, (b3 + bd).to(dev), y, H=Ho, W=Wo, Cin=K1, Cout=Cout, H2=H2, W2=W2, Cin2=K2, stride2=stride,
                             relu=True, batch=B))
    torch.cuda.synchronize()
    assert_close(y, ref.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout), 1e-2, 'conv3 + shortcut')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,W', [(2, 97, 129), (1, 480, 854), (3, 50, 262), (2, 481, 849)])
def test_stem7x7s2_direct(dev, B, H, W, dt):
    """rmem_stem7x7s2 (+ rmem_image_ptrs_to_nhwc4p): the ResNet stem (encoders/resnet.py:131-135, BatchNorm folded) read straight from the
    zero-bordered NHWC4 frames -- against fp32 F.conv2d on the rounded operands, and against the generic row-run GEMM form on the
    8-channel layout (same products, another summation order: equal up to the output rounding).  Odd and even sizes, ragged last
    tile of a row, several images, persistent workgroups that take several tiles each."""
    from rmem_ocu_amd import ops
    imgs = [seeded(70 + b, (3, H, W)).to(dev) for b in range(B)]
    w = seeded(75, (64, 3, 7, 7), 1.0 / 147 ** 0.5).to(dt).float()
    bias = seeded(76, (64,), 0.1)
    x = torch.stack([i.cpu() for i in imgs]).to(dt).float()
    ref = F.relu(F.conv2d(x, w, bias, stride=2, padding=3))
    Ho, Wo = ref.shape[2], ref.shape[3]
    ptrs = torch.tensor([i.data_ptr() for i in imgs], dtype=torch.int64, device=dev)
    hp, wp = ops.stem_padded_size(H, W)
    x4 = torch.zeros(B, hp, wp, 4, dtype=dt, device=dev)
    w4 = torch.zeros(64, 8, 8, 4, dtype=dt)
    w4[:, :7, :7, :3] = w.permute(0, 2, 3, 1).to(dt)
    y = torch.zeros(B, Ho * Wo, 64, dtype=dt, device=dev)
    for _ in range(2):                 # twice: the border must still be zero after the first pass
        ops.run([ops.image_ptrs_to_nhwc4p(ptrs, x4, H=H, W=W, images=B), ops.stem7x7s2(x4, w4.to(dev), bias.to(dev), y, H=H, W=W, images=B)])
    torch.cuda.synchronize()
    assert_close(y.view(B, Ho, Wo, 64), ref.permute(0, 2, 3, 1), 1e-2, 'stem')
    x8 = torch.zeros(B, H * W, 8, dtype=dt, device=dev)
    w8 = F.pad(w.permute(0, 2, 3, 1), (0, 5)).contiguous().to(dt).to(dev)
    y8 = torch.zeros_like(y)
    ops.run([ops.image_ptrs_to_nhwc8(ptrs, x8, H=H, W=W, images=B),
             ops.conv2d(x8, w8, bias.to(dev), y8, H=H, W=W, Cin=8, Cout=64, KH=7, KW=7, stride=2, pad=3, relu=True, batch=B)])
    torch.cuda.synchronize()
    d = (y.float() - y8.float()).abs()
    assert d.max().item() <= 2.0 ** (-7 if dt == torch.bfloat16 else -10) * max(1.0, y8.float().abs().max().item()), d.max().item()
    assert (d > 0).float().mean().item() < 0.02        # a different summation order moves a rounding now and then, no more


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,W', [(2, 97, 129), (1, 480, 854), (3, 50, 262), (2, 481, 849), (1, 33, 15)])
def test_stem_pool_fused(dev, B, H, W, dt):
    """rmem_stem7x7s2_pool: the ResNet stem and its 3x3 stride-2 max-pool in one pass (encoders/resnet.py:131-136; the half-resolution
    map is never written) against rmem_stem7x7s2 followed by rmem_maxpool3x3s2_nhwc: BIT-IDENTICAL, and against fp32 torch.  Odd / even
    sizes, several strips and runs, a last strip narrower than a strip, a map narrower than one strip."""
    from rmem_ocu_amd import ops
    imgs = [seeded(90 + b, (3, H, W)).to(dev) for b in range(B)]
    w = seeded(95, (64, 3, 7, 7), 1.0 / 147 ** 0.5).to(dt).float()
    bias = seeded(96, (64,), 0.1)
    x = torch.stack([i.cpu() for i in imgs]).to(dt).float()
    conv = F.relu(F.conv2d(x, w, bias, stride=2, padding=3)).to(dt).float()
    ref = F.max_pool2d(conv, 3, 2, 1)
    Ho, Wo, Hq, Wq = conv.shape[2], conv.shape[3], ref.shape[2], ref.shape[3]
    ptrs = torch.tensor([i.data_ptr() for i in imgs], dtype=torch.int64, device=dev)
    hp, wp = ops.stem_padded_size(H, W)
    x4 = torch.zeros(B, hp, wp, 4, dtype=dt, device=dev)
    w4 = torch.zeros(64, 8, 8, 4, dtype=dt)
    w4[:, :7, :7, :3] = w.permute(0, 2, 3, 1).to(dt)
    w4 = w4.to(dev)
    y = torch.zeros(B, Ho * Wo, 64, dtype=dt, device=dev)
    p0, p1 = (torch.zeros(B, Hq * Wq, 64, dtype=dt, device=dev) for _ in range(2))
    ops.run([ops.image_ptrs_to_nhwc4p(ptrs, x4, H=H, W=W, images=B), ops.stem7x7s2(x4, w4, bias.to(dev), y, H=H, W=W, images=B),
             ops.maxpool3x3s2(y, p0, H=Ho, W=Wo, C=64, images=B), ops.stem7x7s2_pool(x4, w4, bias.to(dev), p1, H=H, W=W, images=B)])
    torch.cuda.synchronize()
    assert_close(p1.view(B, Hq, Wq, 64), ref.permute(0, 2, 3, 1), 1e-2, 'stem + pool')
    assert torch.equal(p0, p1), f'{(p0 != p1).sum().item()} elements differ from stem7x7s2 + maxpool'


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('C,relu', [(64, True), (128, False), (128, True)])
@pytest.mark.parametrize('B,H,W', [(2, 31, 54), (1, 121, 213), (3, 17, 64), (2, 9, 130), (1, 3, 1)])
def test_conv3x3_direct(dev, B, H, W, C, relu, dt):
    """rmem_conv3x3_direct (3x3 convs with C = 64 / 128 channels read in place from rows kept in LDS, weights in registers: the ResNet
    layer-1 / layer-2 bottlenecks, encoders/resnet.py:52-56, and the decoder's conv_4x, decoders/fpn.py:54-58) against rmem_conv2d_nhwc:
    BIT-IDENTICAL (same k order and epilogue), and against fp32 F.conv2d.  Widths that are a multiple of the 62-pixel strip, ragged,
    and narrower than a strip; image borders; several rows and runs per workgroup; with and without ReLU."""
    from rmem_ocu_amd import ops
    x = seeded(80, (B, C, H, W)).to(dt).float()
    w = seeded(81, (C, C, 3, 3), 1.0 / (3 * C ** 0.5)).to(dt).float()
    bias = seeded(82, (C,), 0.1)
    ref = F.conv2d(x, w, bias, padding=1)
    ref = (F.relu(ref) if relu else ref).permute(0, 2, 3, 1).reshape(B * H * W, C)
    xd = x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(dt).to(dev)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dt).to(dev)
    y0, y1 = (torch.zeros(B * H * W, C, dtype=dt, device=dev) for _ in range(2))
    ops.run(ops.conv2d(xd, wd, bias.to(dev), y0, H=H, W=W, Cin=C, Cout=C, KH=3, KW=3, stride=1, pad=1, relu=relu, batch=B))
    ops.run(ops.conv3x3_direct(xd, wd, bias.to(dev), y1, H=H, W=W, C=C, images=B, relu=relu))
    torch.cuda.synchronize()
    assert_close(y1, ref, 1e-2, 'direct 3x3')
    assert torch.equal(y0, y1), f'{(y0 != y1).sum().item()} elements differ from rmem_conv2d_nhwc'


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('N2', [64, 128])
@pytest.mark.parametrize('dual', [False, True])
def test_bneck_chain_is_bit_identical(dev, dual, N2, dt):
    """rmem_bneck_chain: a bottleneck's conv3 + shortcut + ReLU chained into the next block's conv1 in one launch (the 256-channel tile
    never returns from HBM; encoders/resnet.py:62-68, 48-50) against the launches it replaces -- rmem_conv2d_nhwc with a residual (or
    rmem_conv1x1_dual_nhwc for the block with the strided 1x1 shortcut) followed by rmem_conv2d_nhwc: both outputs bit for bit, with a
    ragged last row tile (M = 2 * 37 * 41 rows), and against fp32 torch on the rounded operands."""
    from rmem_ocu_amd import ops
    B, Ho, Wo, K1, stride = 2, 37, 41, 64, 2 if dual else 1
    H2, W2 = (Ho - 1) * stride + 1, (Wo - 1) * stride + 2
    M = B * Ho * Wo
    rt = lambda t: t.to(dt).float()       # noqa: E731
    b = rt(seeded(61, (M, K1)))
    w3 = rt(seeded(62, (256, K1), 1.0 / K1 ** 0.5))
    b3 = seeded(63, (256,), 0.1)
    w1 = rt(seeded(64, (N2, 256), 1.0 / 16))
    b1 = seeded(65, (N2,), 0.1)
    bd = b.to(dt).to(dev)
    if dual:
        x2 = rt(seeded(66, (B, H2, W2, 64)))
        wd = rt(seeded(67, (256, 64), 1.0 / 8))
        w3d = torch.cat([w3, wd], 1).contiguous().to(dt).to(dev)
        x2d = x2.to(dt).to(dev)
        ref_y = F.relu(b @ w3.t() + x2[:, ::stride, ::stride][:, :Ho, :Wo].reshape(M, 64) @ wd.t() + b3)
    else:
        res = rt(seeded(68, (M, 256)))
        w3d = w3.to(dt).to(dev)
        resd = res.to(dt).to(dev)
        ref_y = F.relu(b @ w3.t() + b3 + res)
    w1d = w1.to(dt).to(dev)
    y0, y1 = (torch.zeros(M, 256, dtype=dt, device=dev) for _ in range(2))
    a0, a1 = (torch.zeros(M, N2, dtype=dt, device=dev) for _ in range(2))
    if dual:
        ops.run(ops.conv1x1_dual(bd, x2d, w3d, b3.to(dev), y0, H=Ho, W=Wo, Cin=K1, Cout=256, H2=H2, W2=W2, Cin2=64, stride2=stride, relu=True, batch=B))
        chain = ops.bneck_chain(bd, w3d, b3.to(dev), y1, w1d, b1.to(dev), a1, H=Ho, W=Wo, K1=K1, N2=N2, x2=x2d, H2=H2, W2=W2, Cin2=64,
                                stride2=stride, batch=B)
    else:
        ops.run(ops.conv2d(bd, w3d, b3.to(dev), y0, H=Ho, W=Wo, Cin=K1, Cout=256, residual=resd, relu=True, batch=B))
        chain = ops.bneck_chain(bd, w3d, b3.to(dev), y1, w1d, b1.to(dev), a1, H=Ho, W=Wo, K1=K1, N2=N2, residual=resd, batch=B)
    ops.run(ops.conv2d(y0, w1d, b1.to(dev), a0, H=Ho, W=Wo, Cin=256, Cout=N2, relu=True, batch=B))
    ops.run(chain)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1), f'y differs in {(y0 != y1).sum().item()} elements'
    assert torch.equal(a0, a1), f'a2 differs in {(a0 != a1).sum().item()} elements'
    assert_close(y1, ref_y, 1e-2, 'chained y')
    assert_close(a1, F.relu(y1.float().cpu() @ w1.t() + b1), 1e-2, 'chained a2')


@pytest.mark.parametrize('align', [True, False])
def test_conv_resized_residual_is_bit_identical(dev, align):
    """rmem_conv_desc.res_up_*: the residual is a lower-resolution map resized on the fly in the GEMM epilogue
    (decoders/fpn.py:49-52: F.interpolate + adapter) -- bit-identical to rmem_bilinear_nhwc followed by a plain residual add,
    and close to the fp32 torch reference; a batch of 2 images, 128x128 and 64x64 tile routes."""
    from rmem_ocu_amd import ops
    for (B, Hi, Wi, Ho, Wo, Cin, Cout) in [(2, 31, 54, 61, 107, 512, 256), (2, 9, 11, 17, 21, 64, 128)]:
        x = rb(seeded(61, (B, Cin, Ho, Wo)))
        lo = rb(seeded(62, (B, Cout, Hi, Wi)))
        w = rb(seeded(63, (Cout, Cin, 1, 1), 1.0 / Cin ** 0.5))
        b = seeded(64, (Cout,), 0.1)
        ref = F.conv2d(x, w, b) + F.interpolate(lo, size=(Ho, Wo), mode='bilinear', align_corners=align)
        xd = x.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cin).contiguous().to(BF16).to(dev)
        lod = lo.permute(0, 2, 3, 1).reshape(B * Hi * Wi, Cout).contiguous().to(BF16).to(dev)
        wd, bd = pack_w(w).to(dev), b.to(dev)
        up = torch.zeros(B * Ho * Wo, Cout, dtype=BF16, device=dev)
        y1 = torch.zeros(B * Ho * Wo, Cout, dtype=BF16, device=dev)
        y2 = torch.zeros(B * Ho * Wo, Cout, dtype=BF16, device=dev)
        ops.run([ops.bilinear(lod, up, Hi=Hi, Wi=Wi, Ho=Ho, Wo=Wo, C=Cout, align_corners=align, images=B),
                 ops.conv2d(xd, wd, bd, y1, H=Ho, W=Wo, Cin=Cin, Cout=Cout, batch=B, residual=up)])
        ops.run(ops.conv2d(xd, wd, bd, y2, H=Ho, W=Wo, Cin=Cin, Cout=Cout, batch=B, residual=lod, res_up=(Hi, Wi, align)))
        torch.cuda.synchronize()
        assert torch.equal(y1, y2), 'fused resize differs from bilinear + residual'
        assert_close(y2, ref.permute(0, 2, 3, 1).reshape(B * Ho * Wo, Cout), 1e-2, 'adapter + resized residual')


@pytest.mark.parametrize('B,M', [(1, 500), (3, 1674 + 37)])
def test_groupnorm_head_fused(dev, B, M):
    """conv_out(relu(gn(x))) (decoders/fpn.py:62-66) as one pass: against the two-launch route (GroupNorm + 1x1 GEMM; same
    bf16 operands, fp32 summation order differs) and against the fp32 torch reference."""
    from rmem_ocu_amd import ops
    C, N = 128, 11
    x = rb(seeded(71, (B, C, M, 1), 2.0))
    g, b = 1 + seeded(72, (C,), 0.1), seeded(73, (C,), 0.1)
    w, wb = rb(seeded(74, (N, C, 1, 1), 1.0 / C ** 0.5)), seeded(75, (N,), 0.1)
    ref = F.conv2d(rb(F.relu(F.group_norm(x, 8, g, b, 1e-5))), w, wb)[:, :, :, 0].permute(0, 2, 1)      # [B, M, N]
    xd = x[:, :, :, 0].permute(0, 2, 1).contiguous().to(BF16).to(dev)
    ws = ops.groupnorm_workspace(8, dev, images=B)
    wd = w.reshape(N, C).contiguous().to(BF16).to(dev)
    y1 = torch.zeros(B * M, 16, dtype=F32, device=dev)
    y2 = torch.zeros(B * M, 16, dtype=F32, device=dev)
    tmp = torch.zeros(B, M, C, dtype=BF16, device=dev)
    ops.run([ops.groupnorm(xd, g.to(dev), b.to(dev), tmp, ws, M=M, C=C, groups=8, act=1, images=B),
             ops.conv2d(tmp, wd, wb.to(dev), y1, H=B * M, W=1, Cin=C, Cout=N, ldo=16)])
    ops.run(ops.groupnorm_head(xd, g.to(dev), b.to(dev), wd, wb.to(dev), y2, ws, M=M, C=C, groups=8, N=N, ldy=16, act=1, images=B))
    torch.cuda.synchronize()
    assert y2[:, N:].abs().max().item() == 0
    assert_close(y2[:, :N], y1[:, :N], 1e-5, 'fused head vs two launches')
    assert_close(y2[:, :N], ref.reshape(B * M, N), 1e-2, 'fused head vs torch')


def test_grouped_launches(dev):
    """rmem_linear_grouped / rmem_add16_grouped / rmem_layernorm256_pair are bit-identical to the single launches."""
    from rmem_ocu_amd import ops
    M, K, N, n = 1674, 256, 256, 3
    xs = [rb(seeded(20 + i, (M, K))).to(BF16).to(dev) for i in range(n)]
    ws = [rb(seeded(30 + i, (N, K), 1 / 16.0)).to(BF16).to(dev) for i in range(n)]
    bs = [seeded(40 + i, (N,), 0.1).to(dev) for i in range(n)]
    ys = [torch.zeros(M, N, dtype=BF16, device=dev) for _ in range(n)]
    y1 = [torch.zeros(M, N, dtype=BF16, device=dev) for _ in range(n)]
    ops.run(ops.linear_grouped(xs, ws, bs, ys, M=M, K=K, N=N))
    ops.run([ops.linear(xs[i], ws[i], bs[i], y1[i], M=M, K=K, N=N) for i in range(n)])
    torch.cuda.synchronize()
    for i in range(n):
        assert torch.equal(ys[i], y1[i])
        assert_close(ys[i], xs[i].float() @ ws[i].float().t() + bs[i], 1e-2, 'grouped linear')
    a = [rb(seeded(50 + i, (M, 256))).to(BF16).to(dev) for i in range(6)]
    out = [torch.zeros(M, 256, dtype=BF16, device=dev) for _ in range(6)]
    ops.run(ops.add16_grouped(a, [a[0]] * 6, out, M * 256))
    torch.cuda.synchronize()
    for i in range(6):
        assert torch.equal(out[i], (a[i].float() + a[0].float()).to(BF16))
    g, be = (1 + seeded(60, (256,), 0.1)).to(dev), seeded(61, (256,), 0.1).to(dev)
    p0, p1, s0, s1 = (torch.zeros(M, 256, dtype=BF16, device=dev) for _ in range(4))
    ops.run(ops.layernorm256_pair(a[0], a[1], p0, a[2], a[3], p1, g, be, M=M))
    ops.run([ops.layernorm256(a[0], g, be, M=M, b=a[1], y=s0), ops.layernorm256(a[2], g, be, M=M, b=a[3], y=s1)])
    torch.cuda.synchronize()
    assert torch.equal(p0, s0) and torch.equal(p1, s1)


@pytest.mark.parametrize('H,W,C,groups,act', [(31, 54, 1024, 32, 2), (11, 13, 1024, 32, 2), (9, 20, 128, 8, 1)])
def test_gn_act_dwconv_fused_is_bit_identical(dev, H, W, C, groups, act):
    """The fused GroupNorm + activation + depth-wise 5x5 equals rmem_groupnorm_nhwc followed by rmem_dwconv5x5_nhwc bit for bit
    (ragged tiles at the right / bottom border included)."""
    from rmem_ocu_amd import ops
    M = H * W
    x = rb(seeded(71, (M, C), 2.0)).to(BF16).to(dev)
    g, b = (1 + seeded(72, (C,), 0.1)).to(dev), seeded(73, (C,), 0.1).to(dev)
    wt = seeded(74, (25, C), 0.2).to(dev)
    ws = ops.groupnorm_workspace(64, dev)
    mid = torch.zeros(M, C, dtype=BF16, device=dev)
    y0 = torch.zeros(M, C, dtype=BF16, device=dev)
    y1 = torch.zeros(M, C, dtype=BF16, device=dev)
    ops.run([ops.groupnorm(x, g, b, mid, ws, M=M, C=C, groups=groups, act=act), ops.dwconv5x5(mid, wt, y0, H=H, W=W, C=C)])
    ops.run(ops.gn_act_dwconv5x5(x, g, b, wt, y1, ws, H=H, W=W, C=C, groups=groups, act=act))
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


def test_batched_forms_match_single_launches(dev):
    """images / clips > 1 (several clips per launch): GroupNorm, the fused GN + dw-conv, bilinear and both attention flavours give
    bit-identical results to one launch per clip."""
    from rmem_ocu_amd import ops
    B, H, W, C = 3, 9, 14, 256
    M = H * W
    x = rb(seeded(81, (B, M, C), 2.0)).to(BF16).to(dev)
    g, b = (1 + seeded(82, (C,), 0.1)).to(dev), seeded(83, (C,), 0.1).to(dev)
    wt = seeded(84, (25, C), 0.2).to(dev)
    ws = ops.groupnorm_workspace(8, dev, images=B)
    yb, y1 = torch.zeros(B, M, C, dtype=BF16, device=dev), torch.zeros(B, M, C, dtype=BF16, device=dev)
    ops.run(ops.groupnorm(x, g, b, yb, ws, M=M, C=C, groups=8, act=1, images=B))
    ops.run([ops.groupnorm(x[i], g, b, y1[i], ws, M=M, C=C, groups=8, act=1) for i in range(B)])
    torch.cuda.synchronize()
    assert torch.equal(yb, y1)
    ops.run(ops.gn_act_dwconv5x5(x, g, b, wt, yb, ws, H=H, W=W, C=C, groups=8, act=2, images=B))
    ops.run([ops.gn_act_dwconv5x5(x[i], g, b, wt, y1[i], ws, H=H, W=W, C=C, groups=8, act=2) for i in range(B)])
    torch.cuda.synchronize()
    assert torch.equal(yb, y1)
    ub, u1 = torch.zeros(B, 17 * 27, C, dtype=BF16, device=dev), torch.zeros(B, 17 * 27, C, dtype=BF16, device=dev)
    ops.run(ops.bilinear(x, ub, Hi=H, Wi=W, Ho=17, Wo=27, C=C, images=B))
    ops.run([ops.bilinear(x[i], u1[i], Hi=H, Wi=W, Ho=17, Wo=27, C=C) for i in range(B)])
    torch.cuda.synchronize()
    assert torch.equal(ub, u1)
    # layout / resampling forms: image conversion, max-pool, label one-hot, logits -> labels
    img = seeded(90, (B, 3, 21, 30), 1.0).to(dev)
    nb, n1 = torch.zeros(B, 21 * 30, 8, dtype=BF16, device=dev), torch.zeros(B, 21 * 30, 8, dtype=BF16, device=dev)
    ops.run(ops.image_to_nhwc8(img, nb, H=21, W=30, images=B))
    ops.run([ops.image_to_nhwc8(img[i], n1[i], H=21, W=30) for i in range(B)])
    pb, p1 = torch.zeros(B, 5 * 7, C, dtype=BF16, device=dev), torch.zeros(B, 5 * 7, C, dtype=BF16, device=dev)
    ops.run(ops.maxpool3x3s2(x, pb, H=H, W=W, C=C, images=B))
    ops.run([ops.maxpool3x3s2(x[i], p1[i], H=H, W=W, C=C) for i in range(B)])
    lab = (seeded(91, (B, 13, 19), 1.0).abs() * 3).clamp(max=4).to(torch.uint8).to(dev)
    hb, h1 = torch.zeros(B, 21 * 30, 16, dtype=BF16, device=dev), torch.zeros(B, 21 * 30, 16, dtype=BF16, device=dev)
    ops.run(ops.label_to_onehot16(lab, hb, Hs=13, Ws=19, Hd=21, Wd=30, ncls=11, images=B))
    ops.run([ops.label_to_onehot16(lab[i], h1[i], Hs=13, Ws=19, Hd=21, Wd=30, ncls=11) for i in range(B)])
    lg = seeded(92, (B, M, 16), 2.0).to(dev)
    lb, l1 = torch.zeros(B, 33, 51, dtype=torch.uint8, device=dev), torch.zeros(B, 33, 51, dtype=torch.uint8, device=dev)
    ops.run(ops.logits_post(lg, ldl=16, nc=11, keep=4, Hi=H, Wi=W, Ho=33, Wo=51, label_u8=lb, images=B))
    ops.run([ops.logits_post(lg[i], ldl=16, nc=11, keep=4, Hi=H, Wi=W, Ho=33, Wo=51, label_u8=l1[i]) for i in range(B)])
    torch.cuda.synchronize()
    assert torch.equal(nb, n1) and torch.equal(pb, p1) and torch.equal(hb, h1) and torch.equal(lb, l1)
    assert n1.float().abs().sum() > 0 and p1.float().abs().sum() > 0 and h1.float().sum() > 0 and l1.sum() > 0
    # attention: one-frame flavour (direct output) and memory read over per-clip banks with different slot orders
    L, T, S = 150, 3, 4
    q = rb(seeded(85, (B, L, 768))).to(BF16).to(dev)
    ob, o1 = torch.zeros(B, L, 256, dtype=BF16, device=dev), torch.zeros(B, L, 256, dtype=BF16, device=dev)
    wsa = ops.attn_workspace(L, 8, 8, dev, nclips=B)
    ops.run(ops.mem_read_attn(q, q.view(-1)[256:], q.view(-1)[512:], ob, wsa, Lq=L, ldq=768, ldkv=768, ldo=256, nchunks=1, lk_single=L,
                              nclips=B, q_cs=L * 768, kv_cs=L * 768, out_cs=L * 256))
    ops.run([ops.mem_read_attn(q[i], q[i].view(-1)[256:], q[i].view(-1)[512:], o1[i], wsa, Lq=L, ldq=768, ldkv=768, ldo=256, nchunks=1,
                               lk_single=L) for i in range(B)])
    torch.cuda.synchronize()
    assert torch.equal(ob, o1)
    kb = rb(seeded(86, (B * S, L, 256))).to(BF16).to(dev)
    vb = rb(seeded(87, (B * S, L, 256))).to(BF16).to(dev)
    pe_cur, pe_mem = seeded(88, (256,), 0.3).to(dev), seeded(89, (4, 256), 0.3).to(dev)
    orders = [[0, 1, 2], [3, 0, 2], [1, 3, 0]]
    rows = [(c * S + orders[c][t], 0, L, t, t) for c in range(B) for t in range(T)]
    tab = ops.make_chunk_table(rows).to(dev)
    mb, m1 = torch.zeros(B, L, T, dtype=F32, device=dev), torch.zeros(B, L, T, dtype=F32, device=dev)
    qq = q[:, :, :256].contiguous()
    ops.run(ops.mem_read_attn(qq, kb, vb, ob, wsa, Lq=L, ldq=256, ldkv=256, ldo=256, slot_stride=L * 256, chunks=tab, nchunks=T,
                              lk_single=T * L, pe_cur=pe_cur, pe_mem=pe_mem, mass=mb, T=T, nclips=B, q_cs=L * 256, out_cs=L * 256))
    for c in range(B):
        tc = ops.make_chunk_table(rows[c * T:(c + 1) * T]).to(dev)
        ops.run(ops.mem_read_attn(qq[c], kb, vb, o1[c], wsa, Lq=L, ldq=256, ldkv=256, ldo=256, slot_stride=L * 256, chunks=tc, nchunks=T,
                                  lk_single=T * L, pe_cur=pe_cur, pe_mem=pe_mem, mass=m1[c], T=T))
    torch.cuda.synchronize()
    assert torch.equal(ob, o1) and torch.equal(mb, m1)


# ---------------------------------------------------------------------------------------------------------------------
# IEEE-half flavour (<name>_f16 entry points; cfg.MODEL_DTYPE = 'fp16'): the same kernels compiled for _Float16 operands.
# Inputs are rounded to half first; with 11 significant bits the 16-bit stores are ~8x finer than bf16's.
F16 = torch.float16


def rh(t):  # round through IEEE half
    return t.to(F16).to(F32)


@pytest.mark.parametrize('case', [(31, 54, 64, 64, 3, 1, 1, True), (61, 107, 256, 512, 1, 2, 0, False), (97, 129, 16, 256, 17, 16, 8, False),
                                  (1674, 1, 1024, 256, 1, 1, 0, False)])
def test_conv2d_fp16(dev, case):
    from rmem_ocu_amd import ops
    H, W, Cin, Cout, k, s, p, relu = case
    x = rh(seeded(900 + H, (1, Cin, H, W)))
    w = rh(seeded(901 + Cout, (Cout, Cin, k, k), (Cin * k * k) ** -0.5))
    b = seeded(902, (Cout,), 0.1)
    ref = F.conv2d(x, w, b, stride=s, padding=p)
    ref = F.relu(ref) if relu else ref
    refm = ref[0].permute(1, 2, 0).reshape(-1, Cout)
    y = torch.zeros(refm.shape[0], Cout, dtype=F16, device=dev)
    xh = x[0].permute(1, 2, 0).reshape(-1, Cin).contiguous().to(F16).to(dev)
    wh = w.permute(0, 2, 3, 1).contiguous().to(F16).to(dev)
    ops.run(ops.conv2d(xh, wh, b.to(dev), y, H=H, W=W, Cin=Cin, Cout=Cout, KH=k, KW=k, stride=s, pad=p, relu=relu))
    torch.cuda.synchronize()
    assert_close(y, refm, 2e-3, f'fp16 conv {case}')


@pytest.mark.parametrize('T,L', [(2, 42), (3, 300), (8, 1674)])
def test_mem_read_attn_fp16(dev, T, L, synth_weights):
    """a1 + a2 + a3 through rmem_mem_read_attn_f16 (always the online-softmax pass: P must stay below 2^16)."""
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.runtime import temporal_slots
    C = 256
    q, k, v = rh(seeded(10 + T, (L, C))), rh(seeded(20 + T, (T, L, C))), rh(seeded(30 + T, (T, L, C)))
    pe_cur, pe_mem = synth_weights['cur_pos_emb'].view(-1), synth_weights['mem_pos_emb']
    slots = temporal_slots(T)
    Qh = ((q + pe_cur) / 32 ** 0.5).view(L, 8, 32).permute(1, 0, 2)
    Kh = (k + pe_mem[slots][:, None, :]).reshape(T * L, 8, 32).permute(1, 2, 0)
    Vh = v.reshape(T * L, 8, 32).permute(1, 0, 2)
    attn = torch.softmax(Qh @ Kh, dim=-1)
    ref = (attn @ Vh).permute(1, 0, 2).reshape(L, C)
    ref_mass = attn.view(8, L, T, L).mean(0).sum(2)
    rows = [(t, 0, L, slots[t], t) for t in range(T)]
    chunks = ops.make_chunk_table(rows).to(dev)
    out = torch.zeros(L, C, dtype=F16, device=dev)
    mass = torch.zeros(L, T, dtype=F32, device=dev)
    ws = ops.attn_workspace(L, 8, T, dev)
    ops.run(ops.mem_read_attn(q.to(F16).to(dev), k.to(F16).to(dev), v.to(F16).to(dev), out, ws, Lq=L, ldq=C, ldkv=C, ldo=C,
                              slot_stride=L * C, chunks=chunks, nchunks=T, pe_cur=pe_cur.to(dev), pe_mem=pe_mem.to(dev).contiguous(),
                              mass=mass, T=T))
    torch.cuda.synchronize()
    # (the scaled query q' = (q + pe) / sqrt(32) * log2 e is rounded to half inside the kernel: ~2^-11 relative on the logits)
    assert_close(out, ref, 4e-3, 'fp16 attention out')
    assert_close(mass, ref_mass, 4e-3, 'fp16 attention mass')


def test_attention_late_dominant_key_fp16(dev):
    """A key 2^100 above everything before it, half flavour: the running maximum must move (P would overflow half at 2^16)."""
    from rmem_ocu_amd import ops
    L, C = 200, 256
    g = torch.Generator().manual_seed(78)
    u = torch.nn.functional.normalize(torch.randn(8, 32, generator=g), dim=1).reshape(C)
    q = rh(seeded(58, (L, C)) * 0.5 + 20.0 * u)
    k = rh(seeded(59, (L, C)) * 0.5)
    v = rh(seeded(60, (L, C)))
    k[150] = rh(20.0 * u)
    Qh = (q / 32 ** 0.5).view(L, 8, 32).permute(1, 0, 2)
    ref = (torch.softmax(Qh @ k.reshape(L, 8, 32).permute(1, 2, 0), -1) @ v.reshape(L, 8, 32).permute(1, 0, 2)).permute(1, 0, 2).reshape(L, C)
    out = torch.zeros(L, C, dtype=F16, device=dev)
    ws = ops.attn_workspace(L, 8, 1, dev)
    ops.run(ops.mem_read_attn(q.to(F16).to(dev), k.to(F16).to(dev), v.to(F16).to(dev), out, ws, Lq=L, ldq=C, ldkv=C, ldo=C,
                              nchunks=1, lk_single=L))
    torch.cuda.synchronize()
    assert_close(out, ref, 4e-3, 'fp16 late dominant key')


def test_norms_fp16(dev):
    """LayerNorm (256) and GroupNorm + GELU + depth-wise 5x5 in the half flavour against fp32 torch."""
    from rmem_ocu_amd import ops
    M, C = 1674, 256
    x = seeded(70, (M, C)) * 2.0 + 0.5
    gam, bet = seeded(71, (C,), 0.5) + 1.0, seeded(72, (C,), 0.1)
    y = torch.zeros(M, C, dtype=F16, device=dev)
    ops.run(ops.layernorm256(x.to(dev), gam.to(dev), bet.to(dev), M=M, y=y))
    torch.cuda.synchronize()
    assert_close(y, F.layer_norm(x, (C,), gam, bet, 1e-5), 2e-3, 'fp16 layernorm')
    H, W, Cf = 31, 54, 1024
    xh = rh(seeded(73, (1, Cf, H, W)))
    g2, b2 = seeded(74, (Cf,), 0.3) + 1.0, seeded(75, (Cf,), 0.1)
    wd = seeded(76, (Cf, 1, 5, 5), 0.2)
    ref = F.conv2d(F.gelu(F.group_norm(xh, 32, g2, b2, 1e-5)), wd, padding=2, groups=Cf)
    yy = torch.zeros(H * W, Cf, dtype=F16, device=dev)
    ws = ops.groupnorm_workspace(32, dev)
    ops.run(ops.gn_act_dwconv5x5(xh[0].permute(1, 2, 0).reshape(-1, Cf).contiguous().to(F16).to(dev), g2.to(dev), b2.to(dev),
                                 wd.reshape(Cf, 25).t().contiguous().to(dev), yy, ws, H=H, W=W, C=Cf, groups=32, act=2))
    torch.cuda.synchronize()
    assert_close(yy, ref[0].permute(1, 2, 0).reshape(-1, Cf), 4e-3, 'fp16 GN + GELU + dwconv5x5')


@pytest.mark.parametrize('L,T,dtype', [(2442, 30, 'bf16'), (3600, 12, 'bf16'), (3600, 12, 'fp16')])
def test_mem_read_full_geometry_properties(dev, synth_weights, L, T, dtype):
    """The memory read at the full geometries of BASELINE cfg 3 / 4 (577x1041 -> HW = 37 x 66 = 2442 tokens, unbounded bank grown
    to T = 30) and cfg 5 (720x1280 -> HW = 45 x 80 = 3600, N = 12; also in the half flavour the config names), through
    size-independent properties -- no oracle at these sizes: per-query mass over the bank frames sums to 1 and is >= 0; a second
    launch is bit-identical; moving every frame to another physical slot changes nothing (bit-identical); the result does not
    depend on how many table rows one workgroup walks; the read is linear in V; a bank of T identical frames reads the same as
    one frame with mass 1/T each."""
    import os
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.runtime import temporal_slots
    E = BF16 if dtype == 'bf16' else F16
    C, S = 256, T + 2
    slots = temporal_slots(T)
    q = seeded(401, (L, C)).to(E).to(dev)
    kb = seeded(402, (S, L, C)).to(E).to(dev)
    vb = seeded(403, (S, L, C)).to(E).to(dev)
    pe_cur = synth_weights['cur_pos_emb'].view(-1).to(dev)
    pe_mem = synth_weights['mem_pos_emb'].to(dev).contiguous()
    ws = ops.attn_workspace(L, 8, T, dev)

    def read(kbank, vbank, order, pe=True):
        out = torch.zeros(L, C, dtype=E, device=dev)
        mass = torch.zeros(L, T, dtype=F32, device=dev)
        tab = ops.make_chunk_table([(int(order[t]), 0, L, slots[t] if pe else -1, t) for t in range(T)]).to(dev)
        ops.run(ops.mem_read_attn(q, kbank, vbank, out, ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=L * C, chunks=tab, nchunks=T,
                                  pe_cur=pe_cur if pe else None, pe_mem=pe_mem if pe else None, mass=mass, T=T))
        torch.cuda.synchronize()
        return out, mass

    order = list(np.random.RandomState(7).permutation(S)[:T])
    out, mass = read(kb, vb, order)
    assert torch.isfinite(out.float()).all() and out.float().abs().mean().item() > 1e-3
    assert (mass.sum(1) - 1.0).abs().max().item() < 1e-4 and mass.min().item() >= 0.0
    out2, mass2 = read(kb, vb, order)
    assert torch.equal(out, out2) and torch.equal(mass, mass2), 'two launches differ'
    order2 = list(np.random.RandomState(8).permutation(S)[:T])
    kb2, vb2 = torch.zeros_like(kb), torch.zeros_like(vb)
    for t in range(T):
        kb2[int(order2[t])] = kb[int(order[t])]
        vb2[int(order2[t])] = vb[int(order[t])]
    outp, massp = read(kb2, vb2, order2)
    assert torch.equal(out, outp) and torch.equal(mass, massp), 'result depends on the physical bank slots'
    res = {}
    for tgt in ('1', '1000000'):                       # all T frames in one workgroup (no partials) / one frame per workgroup
        os.environ['RMEM_ATTN_WGS'] = tgt
        try:
            res[tgt] = read(kb, vb, order)
        finally:
            del os.environ['RMEM_ATTN_WGS']
    tol = 1e-2 if dtype == 'bf16' else 2e-3
    assert_close(res['1'][0], res['1000000'][0].float().cpu(), tol, 'rows per workgroup')
    assert_close(res['1'][0], out.float().cpu(), tol, 'one group vs default grouping')
    assert (res['1'][1] - res['1000000'][1]).abs().max().item() < 1e-4
    # linearity in V
    v2 = seeded(404, (S, L, C)).to(E).to(dev)
    o2, _ = read(kb, v2, order)
    o12, _ = read(kb, (vb.float() + v2.float()).to(E), order)
    assert_close(o12, (out.float() + o2.float()).cpu(), 2e-2 if dtype == 'bf16' else 4e-3, 'linearity in V')
    # T identical frames (no temporal PE): the read equals the one-frame read, every frame gets mass 1 / T
    kb3, vb3 = kb.clone(), vb.clone()
    for t in range(1, T):
        kb3[int(order[t])] = kb[int(order[0])]
        vb3[int(order[t])] = vb[int(order[0])]
    oT, mT = read(kb3, vb3, order, pe=False)
    o1 = torch.zeros(L, C, dtype=E, device=dev)
    ops.run(ops.mem_read_attn(q, kb[int(order[0])], vb[int(order[0])], o1, ws, Lq=L, ldq=C, ldkv=C, ldo=C, nchunks=1, lk_single=L))
    torch.cuda.synchronize()
    assert_close(oT, o1.float().cpu(), tol, 'T identical frames vs one frame')
    assert (mT - 1.0 / T).abs().max().item() < 2e-3 / T


def test_mask_iou_counts_vs_reference_fixture(dev):
    """f4: the device J counts (rmem_mask_iou_counts via evaluator.region_similarity) against J values produced by the reference's
    own evaluation/source/metrics.py::db_eval_iou (tests/golden/iou.npz): several ids, void pixels (label 255), absent ids."""
    import os
    from conftest import GOLDEN
    from rmem_ocu_amd.evaluator import region_similarity
    g = np.load(os.path.join(GOLDEN, 'iou.npz'))
    for i in range(int(g['n'])):
        gt, pred, js, void = g[f'gt{i}'].copy(), g[f'pred{i}'], g[f'j{i}'], g[f'void{i}']
        if void.size:
            gt[void] = 255                        # the DAVIS convention the kernel implements: void pixels carry label 255 in the annotation
        got = region_similarity(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev))
        for k, j in enumerate(js, start=1):
            present = bool(((gt == k) | ((pred == k) & (gt != 255))).any())
            if present:
                assert abs(got[k] - j) < 1e-9, (i, k, got.get(k), j)
            else:
                assert k not in got and j == 1.0   # absent from both maps: the reference defines J = 1, the device skips the id


@pytest.mark.parametrize('dt', ['bf16', 'fp16'])
@pytest.mark.parametrize('hw', [(161, 193), (257, 129)])
def test_lstt_chains_are_bit_identical(dev, dt, hw):
    """The row-block chain kernels (csrc/rowchain.hip: one launch for each row-local sequence of an LSTT block,
    layers/transformer.py:565-576, 635, 659-662, 673-687, 250-259) against the launch list they replace (rmem_conv2d_nhwc,
    rmem_layernorm256, rmem_layernorm256_pair per step): three clips in lockstep through all three blocks from the same random
    state -- the residual stream, curr_Q / curr_V, the norm4 outputs, tgt3, the FFN hidden, the fused QKV and the decoder input must
    come out BIT-IDENTICAL (same MFMA accumulation order, same epilogue order, same LayerNorm reduction, same points of rounding).
    161x193: 143 tokens per clip = 4 full row blocks + one of 15 rows; 257x129: 153 tokens."""
    from rmem_ocu_amd import build_vos_model, get_config, ops
    from rmem_ocu_amd.group_runtime import GroupRuntime
    from rmem_ocu_amd.weights import synth_state_dict
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.MODEL_DTYPE = dt
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    P = model.packed()
    B, T = 3, 2
    outs = []
    for chain in (False, True):
        rt = GroupRuntime(P, hw, 4, dev, B, lookahead=1)
        rt.chain = chain
        rt.chain_stats = False        # (the GroupNorm partial sums out of linear1's epilogue add in another order: next test)
        L = rt.L
        g = torch.Generator().manual_seed(7)
        r = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
        rt.x.copy_(r(B * L, 256))
        for i in range(3):
            rt.short_K[i].copy_(r(B * L, 256)); rt.short_V[i].copy_(r(B * L, 256))
            rt.bank_K[i].copy_(r(*rt.bank_K[i].shape)); rt.bank_V[i].copy_(r(*rt.bank_V[i].shape))
        rt.slots = [[1, 3] for _ in range(B)]
        s = torch.cuda.current_stream().cuda_stream
        rt.prepare_pos(s)
        rt.upload_chunks(s)
        ops.run(rt.prog_lstt(False, T, True), s)
        torch.cuda.synchronize()
        assert len(rt.prog_lstt(False, T, True)) == (19 if chain else 48)
        assert sum(op.name.startswith('rmem_lstt_chain') for op in rt.prog_lstt(False, T, True)) == (10 if chain else 0)
        outs.append({'x': rt.x, 'dec_in': rt.dec_in[:, 256:], 'qkv': rt.qkv, 'h1': rt.h1, 'h3': rt.h3, 'k4': rt.k4, 'v4': rt.v4,
                     'mass': rt.mass[: B * L * T], **{f'cq{i}': rt.curr_Q[i] for i in range(3)}, **{f'cv{i}': rt.curr_V[i] for i in range(3)},
                     **{f'tgt3_{i}': rt.tgt3[i] for i in range(3)}})
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert torch.isfinite(a.float()).all(), k
        assert torch.equal(a, b), f'{k}: {int((a != b).sum())} of {a.numel()} elements differ, max |d| {(a.float() - b.float()).abs().max().item():.3e}'


@pytest.mark.parametrize('geom', [(1674, 8, 4), (2442, 3, 2), (143, 2, 3)])
def test_attn_pair_equals_separate_launches(dev, geom):
    """rmem_lstt_attn_pair_clips (the long-term memory read and the short-term attention of an LSTT block as ONE launch,
    layers/transformer.py:632-635, 656-662) gives bit for bit what rmem_mem_read_attn_clips gives for each of them alone: output,
    second output and the per-frame attention mass, at bench geometry (HW 1674, T = 8, 4 clips: four key groups + merge), at cfg-3
    geometry and on a small ragged case."""
    from rmem_ocu_amd import ops
    L, T, B = geom
    C, S = 256, T + 1
    g = torch.Generator().manual_seed(11)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.7).to(BF16).to(dev)      # noqa: E731
    q, k4, v4 = r(B * L, C), r(B * L, C), r(B * L, C)
    bank_k, bank_v = r(B * S, L, C), r(B * S, L, C)
    pe_cur, pe_mem = torch.randn(C, generator=g).to(dev), torch.randn(4, C, generator=g).to(dev)
    from rmem_ocu_amd.runtime import temporal_slots
    pes = temporal_slots(T)
    splits = max(1, min(8 // T, 32 // T))
    per = (L + splits - 1) // splits
    rows = []
    for c in range(B):
        perm = torch.randperm(S, generator=g)[:T].tolist()           # bank slots in shuffled order, as after evictions
        for t in range(T):
            for kb in range(0, L, per):
                rows.append((c * S + perm[t], kb, min(per, L - kb), pes[t], t))
    n = T * splits
    chunks = ops.make_chunk_table(rows).to(dev)
    ws = ops.attn_workspace(L, 8, 32, dev, nclips=B)
    outs = []
    for pair in (False, True):
        o1, o2 = torch.zeros(B * L, C, dtype=BF16, device=dev), torch.zeros(B * L, C, dtype=BF16, device=dev)
        mass = torch.zeros(B * L * T, dtype=F32, device=dev)
        if pair:
            ops.run(ops.lstt_attn_pair(q, bank_k, bank_v, o1, k4, v4, o2, ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=L * C, chunks=chunks, nchunks=n,
                                       lk_total=T * L, pe_cur=pe_cur, pe_mem=pe_mem, mass=mass, T=T, nclips=B, q_cs=L * C, out_cs=L * C, lk_short=L,
                                       kv_short_cs=L * C, out_short_cs=L * C))
        else:
            ops.run([ops.mem_read_attn(q, bank_k, bank_v, o1, ws, Lq=L, ldq=C, ldkv=C, ldo=C, slot_stride=L * C, chunks=chunks, nchunks=n,
                                       lk_single=T * L, pe_cur=pe_cur, pe_mem=pe_mem, mass=mass, T=T, nclips=B, q_cs=L * C, out_cs=L * C),
                     ops.mem_read_attn(q, k4, v4, o2, ws, Lq=L, ldq=C, ldkv=C, ldo=C, nchunks=1, lk_single=L, nclips=B, q_cs=L * C, kv_cs=L * C,
                                       out_cs=L * C)])
        torch.cuda.synchronize()
        outs.append((o1, o2, mass))
    for a, b, what in zip(outs[0], outs[1], ('memory read', 'short-term attention', 'mass')):
        assert torch.isfinite(a.float()).all() and a.float().abs().max() > 0
        assert torch.equal(a, b), f'{what}: {int((a != b).sum())} of {a.numel()} differ'
    assert (outs[1][2].view(B, L, T).sum(-1) - 1).abs().max().item() < 1e-4


def test_lstt_chain_groupnorm_statistics(dev):
    """Chain B also emits the GroupNorm statistics of the FFN hidden (per 32-row block partial sums out of linear1's epilogue,
    layers/basic.py:27-35) so that GroupNorm + GELU + depth-wise 5x5 is ONE launch without a statistics pass.  The partial sums
    are added in another order than rmem_groupnorm's own statistics kernel: the FFN output agrees to fp32 summation rounding
    (a few e16 ulps on single elements), not bit for bit."""
    from rmem_ocu_amd import build_vos_model, get_config, ops
    from rmem_ocu_amd.group_runtime import GroupRuntime
    from rmem_ocu_amd.weights import synth_state_dict
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    P = model.packed()
    B, T = 3, 2
    outs = []
    for stats in (False, True):
        rt = GroupRuntime(P, (161, 193), 4, dev, B, lookahead=1)
        assert rt.chain and rt.chain_stats
        rt.chain_stats = stats
        L = rt.L
        g = torch.Generator().manual_seed(7)
        r = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
        rt.x.copy_(r(B * L, 256))
        for i in range(3):
            rt.short_K[i].copy_(r(B * L, 256)); rt.short_V[i].copy_(r(B * L, 256))
            rt.bank_K[i].copy_(r(*rt.bank_K[i].shape)); rt.bank_V[i].copy_(r(*rt.bank_V[i].shape))
        rt.slots = [[1, 3] for _ in range(B)]
        s = torch.cuda.current_stream().cuda_stream
        rt.prepare_pos(s)
        rt.upload_chunks(s)
        prog = rt.prog_lstt(False, T, True)
        assert len(prog) == 19 and sum(op.name == 'rmem_gn_act_dwconv5x5_prestats_nhwc' for op in prog) == (3 if stats else 0)
        ops.run(prog, s)
        torch.cuda.synchronize()
        outs.append({'h3': rt.h3.float().clone(), 'x': rt.x.clone(), 'dec_in': rt.dec_in[:, 256:].float().clone()})
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        err = (a - b).abs().max().item() / a.abs().max().item()
        frac = (a != b).float().mean().item()
        print(f'chain statistics, {k}: max rel diff {err:.2e}, {100 * frac:.3f} % of the elements differ')
        # (the last block's buffers: a 1e-7 change of a block's statistics flips single e16 roundings, which the next blocks see as
        # input differences -- single-ulp differences on many elements, never more than an ulp or two of the tensor's scale)
        assert torch.isfinite(b).all() and err < 1e-2, (k, err, frac)


def test_tta_merge_vs_reference_fixture(dev):
    """f3: rmem_tta_merge (un-flip, softmax, mean over augmentations, argmax on the device; managers/evaluator.py:427-441) against
    the reference's own flip_tensor + softmax + mean + argmax on seeded logits (tests/golden/tta.npz: 1, 2, 4 and 3 augmentations,
    mixed flips)."""
    import os
    from conftest import GOLDEN
    from rmem_ocu_amd.evaluator import tta_merge
    g = np.load(os.path.join(GOLDEN, 'tta.npz'))
    for i in range(int(g['n'])):
        lg = torch.from_numpy(g[f'logits{i}']).to(dev)
        flips = [bool(f) for f in g[f'flips{i}']]
        label, label_f, prob = tta_merge([lg[a:a + 1].contiguous() for a in range(len(flips))], flips, want_prob=True)
        torch.cuda.synchronize()
        assert np.abs(prob.cpu().numpy() - g[f'prob{i}']).max() < 1e-6
        ref = g[f'label{i}'][0, 0]
        p2 = np.sort(g[f'prob{i}'][0], axis=0)
        tie = (p2[-1] - p2[-2]) < 1e-6
        assert ((label.cpu().numpy() == ref) | tie).all()
        assert torch.equal(label.float(), label_f[0, 0])


def test_resize_nearest_flip(dev):
    """rmem_resize_nearest_flip_f32 = F.interpolate(flip_tensor(x, 3), size, mode='nearest') (managers/evaluator.py:490-522: flip first,
    then resize), bit for bit (it only moves values), for label maps (down / up, ragged ratios) and 3-plane frames (same size)."""
    from rmem_ocu_amd import ops
    for (c, hs, ws, hd, wd) in [(1, 480, 854, 481, 849), (1, 160, 192, 161, 193), (1, 481, 849, 480, 854), (3, 161, 193, 161, 193), (1, 37, 53, 90, 17)]:
        x = (seeded(200 + hs, (1, c, hs, ws)) * 5).round()
        for fl in (False, True):
            dst = torch.full((1, c, hd, wd), -7.0, dtype=F32, device=dev)
            ops.run(ops.resize_nearest_flip(x.to(dev), dst, flip=fl))
            torch.cuda.synchronize()
            ref = F.interpolate(x.flip(3) if fl else x, size=(hd, wd), mode='nearest')
            assert torch.equal(dst.cpu(), ref), (c, hs, ws, hd, wd, fl)


def test_producer_consumer_forms_are_bit_identical(dev):
    """The loader-wave forms of the 128x128 GEMM tile (k_conv_gemm_dma_pc, default for K >= 512 layers) and of the gated attention's
    P.V kernel keep the one-role kernels' MFMA order: scripts/pc_check.py prints output hashes; the switches are read once per
    process, so each setting runs in a child process (one at a time)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, env in (('one_role', {'RMEM_GEMM_PC': '0', 'RMEM_GP_PC': '0'}), ('default', {}), ('ring3', {'RMEM_GEMM_PC': '3'})):
        e = dict(os.environ)
        e.pop('RMEM_GEMM_PC', None), e.pop('RMEM_GP_PC', None)
        e.update(env)
        r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'pc_check.py')], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = [l for l in r.stdout.splitlines() if l and l[0].isalnum()]
        assert len(outs[name]) == 9, r.stdout
    assert outs['default'] == outs['one_role']
    assert outs['ring3'] == outs['one_role']
