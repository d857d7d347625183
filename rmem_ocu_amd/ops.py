"""Prepared launches of the librmem_hip.so kernels on torch device tensors.

Every function validates shapes/dtypes on the host ONCE and returns an ``Op``: a C
function plus its fully marshalled argument tuple.  ``op(stream)`` only enqueues; a
frame is a list of Ops replayed every frame (or captured into a hipGraph once),
which keeps Python out of the per-frame critical path.  torch is used for device
memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import AttnChunk, ConvDesc, RmemError

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32
E16 = (BF16, F16)     # the two 16-bit element types the library is built for (include/rmem.h: <name> and <name>_f16)


def _fn(name: str, dt):
    """The entry point for operands of element type dt: <name> (bfloat16) or <name>_f16 (IEEE half)."""
    if dt not in E16:
        raise RmemError(f'{name}: 16-bit operands must be bfloat16 or float16, got {dt}')
    return getattr(_lib.lib(), name + ('_f16' if dt == F16 else ''))


def _first16(*ts):
    """Element type of the first 16-bit tensor among ts (bfloat16 if there is none: every flavour then does the same work)."""
    for t in ts:
        if t is not None and t.dtype in E16:
            return t.dtype
    return BF16


class Op:
    __slots__ = ('fn', 'args', 'name', 'keep')

    def __init__(self, fn, args, name, keep=()):
        self.fn, self.args, self.name, self.keep = fn, args, name, keep

    def __call__(self, stream: int):
        rc = self.fn(*self.args, stream)
        if rc:
            raise RmemError(f'{self.name} failed ({rc}): {_lib.lib().rmem_last_error_string().decode()}')


def run(ops, stream: Optional[int] = None):
    """Enqueue one Op or a list of Ops on ``stream`` (default: torch's current stream)."""
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    if isinstance(ops, Op):
        ops(stream)
    else:
        for o in ops:
            o(stream)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RmemError('librmem_hip ops need device tensors (no CPU fallback)')


def conv2d(x, w, bias, y, *, H, W, Cin, Cout, KH=1, KW=1, stride=1, pad=0, residual=None, y2=None, relu=False,
           ldo=None, ldr=None, ld2=None, ws=None, ldx=0, act_begin=0, batch=1, res_up=None) -> Op:
    """y = act(conv(x, w) + bias (+ residual)); w is [Cout, KH, KW, Cin] bf16, x NHWC bf16.
    relu: False/0 none, True/1 ReLU, 2 exact GELU, 3 SiLU (channels >= act_begin); ldx: input row stride of a 1x1 problem.
    res_up=(h, w, align_corners): residual is a bf16 [batch, h, w, ldr] map, bilinearly resized to the output size on the fly."""
    _dev(x, w, bias, y, residual, y2)
    dt = w.dtype
    assert x.dtype == dt and w.dtype == dt and w.is_contiguous()
    assert w.numel() == Cout * KH * KW * Cin, (w.shape, Cout, KH, KW, Cin)
    assert bias is None or (bias.dtype == F32 and bias.numel() == Cout)
    assert y.dtype in (dt, F32) and (y2 is None or y2.dtype == dt)
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    ldo = Cout if ldo is None else ldo
    ldr = Cout if ldr is None else ldr
    ld2 = Cout if ld2 is None else ld2
    assert x.numel() >= (batch * H * W - 1) * (ldx or Cin) + Cin and y.numel() >= (batch * Ho * Wo - 1) * ldo + Cout
    assert batch == 1 or not ldx
    ru = (0, 0, 0) if res_up is None else (int(res_up[0]), int(res_up[1]), int(bool(res_up[2])))
    if res_up is not None:
        assert residual is not None and residual.dtype == dt and residual.numel() >= (batch * ru[0] * ru[1] - 1) * ldr + Cout
    d = ConvDesc(H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad, ldo, ldr, ld2, int(relu), int(y.dtype == F32),
                 int(residual is not None and residual.dtype == F32), ldx, batch, act_begin, *ru)
    if ws is not None:   # split-K workspace: use it only if it is big enough for this problem
        _dev(ws)
        if ws.numel() * ws.element_size() < _lib.lib().rmem_conv_workspace_bytes(C.byref(d)):
            ws = None
    args = (C.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(residual), _ptr(y), _ptr(y2), _ptr(ws))
    return Op(_fn('rmem_conv2d_nhwc', dt), args, 'rmem_conv2d_nhwc', (d, x, w, bias, residual, y, y2, ws))


def conv1x1_dual(x, x2, w_cat, bias, y, *, H, W, Cin, Cout, H2, W2, Cin2, stride2, relu=False, batch=1) -> Op:
    """y = act([x | x2 sampled at stride2] @ w_cat^T + bias): bottleneck conv3 + its 1x1 shortcut as one GEMM.
    x [batch*H*W, Cin] bf16, x2 NHWC [batch, H2, W2, Cin2] bf16, w_cat [Cout, Cin + Cin2] bf16."""
    _dev(x, x2, w_cat, bias, y)
    dt = w_cat.dtype
    assert x.dtype == dt and x2.dtype == dt and w_cat.dtype == dt and w_cat.is_contiguous() and y.dtype == dt
    assert w_cat.numel() == Cout * (Cin + Cin2) and bias.dtype == F32 and bias.numel() == Cout
    assert x.numel() >= batch * H * W * Cin and x2.numel() >= batch * H2 * W2 * Cin2 and y.numel() >= batch * H * W * Cout
    d = ConvDesc(H, W, Cin, H, W, Cout, 1, 1, 1, 0, Cout, Cout, Cout, int(relu), 0, 0, 0, batch, 0, 0, 0, 0)
    args = (C.byref(d), _ptr(x), _ptr(x2), H2, W2, Cin2, stride2, _ptr(w_cat), _ptr(bias), _ptr(y))
    return Op(_fn('rmem_conv1x1_dual_nhwc', dt), args, 'rmem_conv1x1_dual_nhwc', (d, x, x2, w_cat, bias, y))


def bneck_chain(b, w3, bias3, y, w1, bias1, a2, *, H, W, K1, N2, residual=None, x2=None, H2=0, W2=0, Cin2=0, stride2=1, batch=1) -> Op:
    """y = relu(b @ w3^T + bias3 + residual) and a2 = relu(y @ w1^T + bias1) in one launch (bottleneck tail + the next block's conv1);
    with x2 instead of residual: y = relu([b | x2 sampled at stride2] @ w3^T + bias3).  b [batch*H*W, K1], y [.., 256], a2 [.., N2]."""
    _dev(b, w3, bias3, y, w1, bias1, a2, residual, x2)
    dt = w3.dtype
    M = batch * H * W
    assert (residual is None) != (x2 is None)
    assert all(t.dtype == dt for t in (b, y, w1, a2)) and bias3.dtype == F32 and bias1.dtype == F32
    assert w3.is_contiguous() and w1.is_contiguous() and w3.numel() == 256 * (K1 + (Cin2 if x2 is not None else 0)) and w1.numel() == N2 * 256
    assert b.numel() >= M * K1 and y.numel() >= M * 256 and a2.numel() >= M * N2 and bias3.numel() == 256 and bias1.numel() == N2
    assert residual is None or (residual.dtype == dt and residual.numel() >= M * 256)
    assert x2 is None or (x2.dtype == dt and x2.numel() >= batch * H2 * W2 * Cin2)
    d = _lib.BneckChainDesc(batch, H, W, K1, 256, N2, H2, W2, Cin2, stride2)
    args = (C.byref(d), _ptr(b), _ptr(x2), _ptr(w3), _ptr(bias3), _ptr(residual), _ptr(y), _ptr(w1), _ptr(bias1), _ptr(a2))
    return Op(_fn('rmem_bneck_chain', dt), args, 'rmem_bneck_chain', (d, b, x2, w3, bias3, residual, y, w1, bias1, a2))


def linear(x, w, bias, y, *, M, K, N, residual=None, y2=None, relu=False, ldo=None, ldr=None, ld2=None, ws=None, ldx=0,
           act_begin=0) -> Op:
    """y[M, N] = x[M, K] @ w[N, K]^T + bias: the 1x1 case of conv2d (x rows of stride ldx, default K)."""
    return conv2d(x, w, bias, y, H=M, W=1, Cin=K, Cout=N, residual=residual, y2=y2, relu=relu, ldo=ldo, ldr=ldr, ld2=ld2, ws=ws,
                  ldx=ldx, act_begin=act_begin)


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[_ptr(t) for t in ts])


def linear_grouped(xs, ws, biases, ys, *, M, K, N, residuals=None, relu=False) -> Op:
    """len(xs) <= 4 GEMMs y_i[M, N] = x_i[M, K] @ w_i[N, K]^T + bias_i (+ residual_i) of identical shape as ONE launch."""
    n = len(xs)
    assert 1 <= n <= 4 and len(ws) == n and len(ys) == n and (biases is None or len(biases) == n)
    _dev(*xs, *ws, *ys, *(biases or ()), *(residuals or ()))
    dt = ws[0].dtype
    for x, w, y in zip(xs, ws, ys):
        assert x.dtype == dt and w.dtype == dt and w.numel() == N * K and x.numel() >= M * K and y.numel() >= M * N
    assert len({y.dtype for y in ys}) == 1 and (residuals is None or len({r.dtype for r in residuals}) == 1)
    d = ConvDesc(M, 1, K, M, 1, N, 1, 1, 1, 0, N, N, N, int(relu), int(ys[0].dtype == F32),
                 int(residuals is not None and residuals[0].dtype == F32), 0, 1, 0)
    arrs = (_ptr_array(xs), _ptr_array(ws), _ptr_array(biases) if biases is not None else None,
            _ptr_array(residuals) if residuals is not None else None, _ptr_array(ys))
    args = (C.byref(d), n, arrs[0], arrs[1], arrs[2], arrs[3], arrs[4])
    return Op(_fn('rmem_linear_grouped', dt), args, 'rmem_linear_grouped', (d, arrs, xs, ws, biases, residuals, ys))


def add16_grouped(as_, bs, ys, n: int) -> Op:
    """len(as_) <= 8 adds y_i = a_i + b_i over n elements each as one launch."""
    _dev(*as_, *bs, *ys)
    dt = as_[0].dtype
    assert len(as_) == len(bs) == len(ys) <= 8 and all(t.dtype == dt for t in (*as_, *bs, *ys))
    arrs = (_ptr_array(as_), _ptr_array(bs), _ptr_array(ys))
    return Op(_fn('rmem_add16_grouped', dt), (len(as_), arrs[0], arrs[1], arrs[2], n), 'rmem_add16_grouped', (arrs, as_, bs, ys))


def layernorm256_pair(a0, b0, y0, a1, b1, y1, gamma, beta, *, M, eps=1e-5) -> Op:
    """y0 = LN(a0 + b0), y1 = LN(a1 + b1) with one weight set, [M, 256] bf16 contiguous, one launch."""
    _dev(a0, b0, y0, a1, b1, y1, gamma, beta)
    dt = a0.dtype
    assert all(t.dtype == dt and t.numel() >= M * 256 for t in (a0, b0, y0, a1, b1, y1)) and gamma.dtype == F32
    args = (_ptr(a0), _ptr(b0), _ptr(y0), _ptr(a1), _ptr(b1), _ptr(y1), _ptr(gamma), _ptr(beta), eps, M)
    return Op(_fn('rmem_layernorm256_pair', dt), args, 'rmem_layernorm256_pair', (a0, b0, y0, a1, b1, y1, gamma, beta))


def _chain(name, cls, ints, ptrs, dt):
    """An rmem_lstt_chain_* launch: ints = (L, clips, eps, third int), ptrs = {field: tensor or None} in the struct's order."""
    ts = list(ptrs.values())
    _dev(*ts)
    for k, t in ptrs.items():
        if t is not None and t.data_ptr() % 16:
            raise RmemError(f'{name}: {k} must be 16-byte aligned')
    d = cls(int(ints[0]), int(ints[1]), float(ints[2]), int(ints[3]), *[_ptr(t) for t in ts])
    return Op(_fn(name, dt), (C.byref(d),), name, (d, ts))


def lstt_chain_a(*, L, clips, att, x, w_proj, b_proj, ln2, curr_v, w_q, b_q, curr_q, short_k, short_v, ln4, k4, v4, eps=1e-5) -> Op:
    """x += att @ Wp^T + bp; curr_v = LN2(x); curr_q = curr_v @ Wq^T + bq; k4 = LN4(short_k + curr_q); v4 = LN4(short_v + curr_v)
    over [clips * L, 256] rows (weights in pack.pack_frag order; ln2 / ln4 = (gamma, beta))."""
    dt = att.dtype
    R = clips * L
    assert x.dtype == F32 and x.numel() >= R * 256 and all(t.dtype == dt and t.numel() >= R * 256 for t in (att, curr_v, curr_q, short_k, short_v, k4, v4))
    assert w_proj.dtype == dt and w_q.dtype == dt and w_proj.numel() == 65536 and w_q.numel() == 65536
    return _chain('rmem_lstt_chain_a', _lib.ChainA, (L, clips, eps, 0),
                  dict(att=att, x=x, w_proj=w_proj, b_proj=b_proj, ln2_g=ln2[0], ln2_b=ln2[1], curr_v=curr_v, w_q=w_q, b_q=b_q, curr_q=curr_q,
                       short_k=short_k, short_v=short_v, ln4_g=ln4[0], ln4_b=ln4[1], k4=k4, v4=v4), dt)


def lstt_chain_b(*, L, clips, att_long, att_short, x, w_long, b_long, w_short, b_short, tgt3, ln3, w1, b1, h1, gn_partial=None,
                 gn_splits=0, eps=1e-5) -> Op:
    """x += att_long @ Wl^T + bl; tgt3 = att_short @ Ws^T + bs; x += tgt3; h1 = LN3(x) @ W1^T + b1 ([rows, 1024])
    (+ per-row-block GroupNorm partial sums of h1: fp32 [clips, 32, gn_splits, 2])."""
    dt = att_long.dtype
    R = clips * L
    assert x.dtype == F32 and all(t.dtype == dt and t.numel() >= R * 256 for t in (att_long, att_short, tgt3)) and h1.dtype == dt and h1.numel() >= R * 1024
    assert w_long.numel() == 65536 and w_short.numel() == 65536 and w1.numel() == 262144 and b1.numel() == 1024
    assert gn_partial is None or (gn_partial.dtype == F32 and gn_splits * 32 >= L and gn_partial.numel() >= clips * 32 * gn_splits * 2)
    return _chain('rmem_lstt_chain_b', _lib.ChainB, (L, clips, eps, gn_splits),
                  dict(att_long=att_long, att_short=att_short, x=x, w_long=w_long, b_long=b_long, w_short=w_short, b_short=b_short, tgt3=tgt3,
                       ln3_g=ln3[0], ln3_b=ln3[1], w1=w1, b1=b1, h1=h1, gn_partial=gn_partial), dt)


def lstt_chain_c(*, L, clips, x, dt, h3=None, w2=None, b2=None, dec_norm=(None, None), dec_out=None, ld_dec=0, ln1=(None, None), w_qkv=None,
                 b_qkv=None, pos_qk=None, qkv=None, eps=1e-5) -> Op:
    """(h3 given) x += h3 @ W2^T + b2; dec_out[:, :256] (row stride ld_dec) = LN_dec(x);  (w_qkv given) qkv = LN1'(x) @ Wqkv^T + b_qkv + pos_qk."""
    R = clips * L
    assert x.dtype == F32 and x.numel() >= R * 256
    assert h3 is None or (h3.dtype == dt and h3.numel() >= R * 1024 and w2.numel() == 262144 and dec_out.numel() >= (R - 1) * ld_dec + 256)
    assert w_qkv is None or (w_qkv.numel() == 196608 and pos_qk.dtype == F32 and pos_qk.numel() >= R * 768 and qkv.dtype == dt and qkv.numel() >= R * 768)
    return _chain('rmem_lstt_chain_c', _lib.ChainC, (L, clips, eps, ld_dec),
                  dict(x=x, h3=h3, w2=w2, b2=b2, dec_g=dec_norm[0], dec_b=dec_norm[1], dec_out=dec_out, ln1_g=ln1[0], ln1_b=ln1[1], w_qkv=w_qkv,
                       b_qkv=b_qkv, pos_qk=pos_qk, qkv=qkv), dt)


def attn_workspace(Lq: int, heads: int, nchunks: int, device, nclips: int = 1) -> torch.Tensor:
    n = nclips * _lib.lib().rmem_attn_workspace_bytes(Lq, heads, nchunks)
    return torch.empty(n // 4, dtype=F32, device=device)


def make_chunk_table(chunks: Sequence[Sequence[int]]) -> torch.Tensor:
    """Host int32 [n, 8] table from (slot, key_begin, key_count, pe_slot, t) rows."""
    t = torch.zeros(len(chunks), 8, dtype=torch.int32)
    for i, c in enumerate(chunks):
        t[i, :5] = torch.tensor(list(c), dtype=torch.int32)
    return t


def mem_read_attn(q, k_bank, v_bank, out, workspace, *, Lq, heads=8, ldq, ldkv, ldo, slot_stride=0, chunks=None,
                  nchunks=1, lk_single=0, pe_cur=None, pe_mem=None, mass=None, T=0, nclips=1, q_cs=0, kv_cs=0, out_cs=0) -> Op:
    """nclips > 1: that many clips of identical shape in one launch (clip c's operands c * {q,kv,out}_cs elements further, its
    chunk rows at chunks[c * nchunks:], its mass at mass[c * Lq * T:])."""
    _dev(q, k_bank, v_bank, out, workspace, chunks, pe_cur, pe_mem, mass)
    dt = q.dtype
    assert q.dtype == dt and k_bank.dtype == dt and v_bank.dtype == dt and out.dtype == dt
    assert workspace.numel() * 4 >= nclips * _lib.lib().rmem_attn_workspace_bytes(Lq, heads, nchunks)
    assert chunks is None or (chunks.dtype == torch.int32 and chunks.numel() >= nclips * nchunks * 8)
    assert mass is None or (mass.dtype == F32 and mass.numel() >= nclips * Lq * T)
    args = (_ptr(q), ldq, _ptr(k_bank), _ptr(v_bank), slot_stride, ldkv, _ptr(chunks), nchunks, lk_single,
            _ptr(pe_cur), _ptr(pe_mem), Lq, heads, _ptr(out), ldo, _ptr(mass), T, nclips, q_cs, kv_cs, out_cs, _ptr(workspace))
    return Op(_fn('rmem_mem_read_attn_clips', dt), args, 'rmem_mem_read_attn',
              (q, k_bank, v_bank, out, workspace, chunks, pe_cur, pe_mem, mass))


def lstt_attn_pair(q, k_bank, v_bank, out_long, k_short, v_short, out_short, workspace, *, Lq, heads=8, ldq, ldkv, ldo, slot_stride, chunks,
                   nchunks, lk_total, pe_cur, pe_mem, mass=None, T=0, nclips=1, q_cs=0, out_cs=0, lk_short, kv_short_cs=0, out_short_cs=0) -> Op:
    """The long-term memory read (mem_read_attn with a chunk table) and the short-term attention of the same queries as one launch."""
    _dev(q, k_bank, v_bank, out_long, k_short, v_short, out_short, workspace, chunks, pe_cur, pe_mem, mass)
    dt = q.dtype
    assert all(t.dtype == dt for t in (k_bank, v_bank, out_long, k_short, v_short, out_short))
    assert workspace.numel() * 4 >= nclips * _lib.lib().rmem_attn_workspace_bytes(Lq, heads, nchunks)
    assert chunks.dtype == torch.int32 and chunks.numel() >= nclips * nchunks * 8
    assert mass is None or (mass.dtype == F32 and mass.numel() >= nclips * Lq * T)
    assert k_short.numel() >= (nclips - 1) * kv_short_cs + (lk_short - 1) * ldkv + heads * 32
    args = (_ptr(q), ldq, _ptr(k_bank), _ptr(v_bank), slot_stride, ldkv, _ptr(chunks), nchunks, lk_total, _ptr(pe_cur), _ptr(pe_mem), Lq, heads,
            _ptr(out_long), ldo, _ptr(mass), T, nclips, q_cs, out_cs, _ptr(k_short), _ptr(v_short), lk_short, kv_short_cs, _ptr(out_short),
            out_short_cs, _ptr(workspace))
    return Op(_fn('rmem_lstt_attn_pair_clips', dt), args, 'rmem_lstt_attn_pair',
              (q, k_bank, v_bank, out_long, k_short, v_short, out_short, workspace, chunks, pe_cur, pe_mem, mass))


def layernorm256(a, gamma, beta, *, M, lda=256, b=None, ldb=256, y=None, ldy=256, pos=None, ypos=None, ldyp=256,
                 yf=None, ldyf=256, eps=1e-5) -> Op:
    _dev(a, b, gamma, beta, y, pos, ypos, yf)
    dt = _first16(y, ypos, a, b)
    assert a.dtype in (dt, F32) and (b is None or b.dtype in (dt, F32))
    assert gamma.dtype == F32 and beta.dtype == F32 and gamma.numel() == 256
    assert (y is None or y.dtype == dt) and (ypos is None or ypos.dtype == dt) and (yf is None or yf.dtype == F32)
    args = (_ptr(a), int(a.dtype == F32), lda, _ptr(b), int(b is not None and b.dtype == F32), ldb, _ptr(gamma), _ptr(beta),
            eps, M, _ptr(y), ldy, _ptr(pos), _ptr(ypos), ldyp, _ptr(yf), ldyf)
    return Op(_fn('rmem_layernorm256', dt), args, 'rmem_layernorm256', (a, b, gamma, beta, y, pos, ypos, yf))


def layernorm(a, gamma, beta, *, M, C, lda=None, y=None, ldy=None, yf=None, ldyf=None, eps=1e-5) -> Op:
    """LayerNorm over C in {128, 256, 512, 1024}."""
    _dev(a, gamma, beta, y, yf)
    dt = _first16(y, a)
    assert a.dtype in (dt, F32) and gamma.dtype == F32 and gamma.numel() == C
    args = (_ptr(a), int(a.dtype == F32), C if lda is None else lda, _ptr(gamma), _ptr(beta), eps, M, C, _ptr(y),
            C if ldy is None else ldy, _ptr(yf), C if ldyf is None else ldyf)
    return Op(_fn('rmem_layernorm', dt), args, 'rmem_layernorm', (a, gamma, beta, y, yf))


def patch_merge_ln(x, gamma, beta, y, *, H, W, C, eps=1e-5, images=1) -> Op:
    """images > 1: x fp32 [images][H][W][C] -> y [images][ceil(H/2) * ceil(W/2)][4C]."""
    _dev(x, gamma, beta, y)
    dt = y.dtype
    assert x.dtype == F32 and y.dtype == dt and gamma.numel() == 4 * C
    return Op(_fn('rmem_patch_merge_ln_images', dt), (_ptr(x), images, H, W, C, _ptr(gamma), _ptr(beta), eps, _ptr(y)), 'rmem_patch_merge_ln',
              (x, gamma, beta, y))


def window_attn(qkv, qkv_bias, table, out, *, H, W, C, heads, shift, images=1) -> Op:
    """images > 1: qkv [images][H * W][3C], out [images][H * W][C] (windows never cross images)."""
    _dev(qkv, qkv_bias, table, out)
    dt = qkv.dtype
    assert qkv.dtype == dt and out.dtype == dt and qkv_bias.dtype == F32 and table.dtype == F32
    assert qkv_bias.numel() == 3 * C and table.numel() == 4 * heads * 49 * 49 and table.is_contiguous()
    assert qkv.numel() >= images * H * W * 3 * C and out.numel() >= images * H * W * C
    return Op(_fn('rmem_window_attn_images', dt), (_ptr(qkv), _ptr(qkv_bias), _ptr(table), _ptr(out), images, H, W, C, heads, shift),
              'rmem_window_attn', (qkv, qkv_bias, table, out))


def add16(a, b, y, n: int) -> Op:
    _dev(a, b, y)
    dt = a.dtype
    assert a.dtype == dt and b.dtype == dt and y.dtype == dt
    return Op(_fn('rmem_add16', dt), (_ptr(a), _ptr(b), _ptr(y), n), 'rmem_add16', (a, b, y))


def groupnorm_workspace(groups: int, device, images: int = 1) -> torch.Tensor:
    return torch.empty(images * _lib.lib().rmem_groupnorm_workspace_bytes(groups) // 4, dtype=F32, device=device)


def groupnorm(x, gamma, beta, y, ws, *, M, C, groups, act=0, eps=1e-5, images=1) -> Op:
    """images > 1: x / y are [images][M][C], statistics per image."""
    _dev(x, gamma, beta, y, ws)
    dt = y.dtype
    assert x.dtype in (dt, F32) and y.dtype == dt and gamma.dtype == F32 and gamma.numel() == C
    assert ws.numel() * 4 >= images * _lib.lib().rmem_groupnorm_workspace_bytes(groups)
    if images > 1:
        assert x.dtype == dt
        args = (_ptr(x), images, M, C, groups, _ptr(gamma), _ptr(beta), eps, act, _ptr(y), _ptr(ws))
        return Op(_fn('rmem_groupnorm_nhwc_images', dt), args, 'rmem_groupnorm_nhwc', (x, gamma, beta, y, ws))
    args = (_ptr(x), M, C, groups, _ptr(gamma), _ptr(beta), eps, act, _ptr(y), _ptr(ws))
    if x.dtype == F32:
        return Op(_fn('rmem_groupnorm_f32_nhwc', dt), args, 'rmem_groupnorm_f32_nhwc', (x, gamma, beta, y, ws))
    return Op(_fn('rmem_groupnorm_nhwc', dt), args, 'rmem_groupnorm_nhwc', (x, gamma, beta, y, ws))


def groupnorm_head(x, gamma, beta, w, bias, y, ws, *, M, C, groups, N, ldy, act=1, eps=1e-5, images=1) -> Op:
    """y[images*M, ldy] fp32: y[:, :N] = act(GroupNorm(x)) @ w^T + bias with w [N, C] bf16 (N <= 16, C == 128), one pass over x."""
    _dev(x, gamma, beta, w, bias, y, ws)
    dt = x.dtype
    assert x.dtype == dt and w.dtype == dt and y.dtype == F32 and gamma.dtype == F32 and gamma.numel() == C
    assert w.numel() == N * C and w.is_contiguous() and (bias is None or (bias.dtype == F32 and bias.numel() == N))
    assert x.numel() >= images * M * C and y.numel() >= (images * M - 1) * ldy + N
    assert ws.numel() * 4 >= images * _lib.lib().rmem_groupnorm_workspace_bytes(groups)
    args = (_ptr(x), images, M, C, groups, _ptr(gamma), _ptr(beta), eps, act, _ptr(w), _ptr(bias), N, _ptr(y), ldy, _ptr(ws))
    return Op(_fn('rmem_groupnorm_head_nhwc_images', dt), args, 'rmem_groupnorm_head_nhwc', (x, gamma, beta, w, bias, y, ws))


def gn_act_dwconv5x5(x, gamma, beta, w_t, y, ws, *, H, W, C, groups, act=2, eps=1e-5, images=1) -> Op:
    """y = dwconv5x5(act(GroupNorm(x))): statistics launch + one fused normalise / activate / convolve launch."""
    _dev(x, gamma, beta, w_t, y, ws)
    dt = x.dtype
    assert x.dtype == dt and y.dtype == dt and w_t.dtype == F32 and w_t.numel() == 25 * C and gamma.numel() == C
    assert ws.numel() * 4 >= images * _lib.lib().rmem_groupnorm_workspace_bytes(groups)
    args = (_ptr(x), images, H, W, C, groups, _ptr(gamma), _ptr(beta), eps, act, _ptr(w_t), _ptr(y), _ptr(ws))
    return Op(_fn('rmem_gn_act_dwconv5x5_nhwc_images', dt), args, 'rmem_gn_act_dwconv5x5_nhwc', (x, gamma, beta, w_t, y, ws))


def gn_act_dwconv5x5_prestats(x, gamma, beta, w_t, y, stats, *, H, W, C, groups, act=2, eps=1e-5, images=1) -> Op:
    """gn_act_dwconv5x5 whose statistics partials (fp32 [images, groups, 64, 2]) were written by x's producer: one launch."""
    _dev(x, gamma, beta, w_t, y, stats)
    dt = x.dtype
    assert x.dtype == dt and y.dtype == dt and w_t.dtype == F32 and w_t.numel() == 25 * C and gamma.numel() == C
    assert stats.dtype == F32 and stats.numel() >= images * groups * 64 * 2
    args = (_ptr(x), images, H, W, C, groups, _ptr(gamma), _ptr(beta), eps, act, _ptr(w_t), _ptr(y), _ptr(stats))
    return Op(_fn('rmem_gn_act_dwconv5x5_prestats_nhwc_images', dt), args, 'rmem_gn_act_dwconv5x5_prestats_nhwc', (x, gamma, beta, w_t, y, stats))


def dwconv5x5(x, w_t, y, *, H, W, C) -> Op:
    _dev(x, w_t, y)
    dt = x.dtype
    assert x.dtype == dt and y.dtype == dt and w_t.dtype == F32 and w_t.numel() == 25 * C
    return Op(_fn('rmem_dwconv5x5_nhwc', dt), (_ptr(x), _ptr(w_t), _ptr(y), H, W, C), 'rmem_dwconv5x5_nhwc', (x, w_t, y))


def image_to_nhwc8(img, out, *, H, W, images=1) -> Op:
    """fp32 [images, 3, H, W] -> bf16 [images, H*W, 8]"""
    _dev(img, out)
    dt = out.dtype
    assert img.dtype == F32 and img.is_contiguous() and img.numel() == images * 3 * H * W and out.dtype == dt
    assert out.is_contiguous() and out.numel() >= images * H * W * 8
    return Op(_fn('rmem_image_to_nhwc8_images', dt), (_ptr(img), _ptr(out), images, H, W), 'rmem_image_to_nhwc8', (img, out))


def image_ptrs_to_nhwc8(ptr_table, out, *, H, W, images) -> Op:
    """device int64 table of `images` pointers to fp32 [3, H, W] frames -> 16-bit [images, H*W, 8]"""
    _dev(ptr_table, out)
    dt = out.dtype
    assert ptr_table.dtype == torch.int64 and ptr_table.numel() >= images and out.is_contiguous() and out.numel() >= images * H * W * 8
    return Op(_fn('rmem_image_ptrs_to_nhwc8', dt), (_ptr(ptr_table), _ptr(out), images, H, W), 'rmem_image_ptrs_to_nhwc8', (ptr_table, out))


def stem_padded_size(H: int, W: int):
    """(Hp, Wp) of the zero-bordered NHWC4 frame layout rmem_stem7x7s2 reads"""
    hp, wp = C.c_int(0), C.c_int(0)
    if _lib.lib().rmem_stem_padded_size(H, W, C.byref(hp), C.byref(wp)):
        raise RmemError('rmem_stem_padded_size: bad argument')
    return hp.value, wp.value


def image_ptrs_to_nhwc4p(ptr_table, out, *, H, W, images) -> Op:
    """device int64 table of `images` pointers to fp32 [3, H, W] frames -> the interior of 16-bit [images, Hp, Wp, 4] (border kept zero)"""
    _dev(ptr_table, out)
    dt = out.dtype
    hp, wp = stem_padded_size(H, W)
    assert ptr_table.dtype == torch.int64 and ptr_table.numel() >= images and out.is_contiguous() and out.numel() >= images * hp * wp * 4
    return Op(_fn('rmem_image_ptrs_to_nhwc4p', dt), (_ptr(ptr_table), _ptr(out), images, H, W), 'rmem_image_ptrs_to_nhwc4p', (ptr_table, out))


def stem7x7s2(x_padded, w4, bias, y, *, H, W, images) -> Op:
    """y [images, Ho*Wo, 64] = relu(conv7x7 stride 2 of the padded NHWC4 frames + bias); w4 [64, 8, 8, 4]"""
    _dev(x_padded, w4, bias, y)
    dt = w4.dtype
    hp, wp = stem_padded_size(H, W)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert x_padded.dtype == dt and y.dtype == dt and bias.dtype == F32 and w4.is_contiguous() and w4.numel() == 64 * 256 and bias.numel() == 64
    assert x_padded.numel() >= images * hp * wp * 4 and y.numel() >= images * ho * wo * 64
    return Op(_fn('rmem_stem7x7s2', dt), (_ptr(x_padded), images, H, W, _ptr(w4), _ptr(bias), _ptr(y)), 'rmem_stem7x7s2', (x_padded, w4, bias, y))


def conv3x3_c64_direct(x, w, bias, y, *, H, W, images) -> Op:
    """y = relu(conv3x3(x) + bias), 64 -> 64 channels, stride 1, pad 1; x / y [images*H*W, 64], w [64, 3, 3, 64]"""
    _dev(x, w, bias, y)
    dt = w.dtype
    assert x.dtype == dt and y.dtype == dt and bias.dtype == F32 and w.is_contiguous() and w.numel() == 64 * 576 and bias.numel() == 64
    assert x.numel() >= images * H * W * 64 and y.numel() >= images * H * W * 64
    return Op(_fn('rmem_conv3x3_c64_direct', dt), (_ptr(x), images, H, W, _ptr(w), _ptr(bias), _ptr(y)), 'rmem_conv3x3_c64_direct', (x, w, bias, y))


def stem7x7s2_pool(x_padded, w4, bias, y_pooled, *, H, W, images) -> Op:
    """y_pooled [images, HP*WP, 64] = maxpool3x3s2(relu(conv7x7 stride 2 of the padded NHWC4 frames + bias)) in one pass"""
    _dev(x_padded, w4, bias, y_pooled)
    dt = w4.dtype
    hp, wp = stem_padded_size(H, W)
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    hq, wq = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
    assert x_padded.dtype == dt and y_pooled.dtype == dt and bias.dtype == F32 and w4.is_contiguous() and w4.numel() == 64 * 256 and bias.numel() == 64
    assert x_padded.numel() >= images * hp * wp * 4 and y_pooled.numel() >= images * hq * wq * 64
    return Op(_fn('rmem_stem7x7s2_pool', dt), (_ptr(x_padded), images, H, W, _ptr(w4), _ptr(bias), _ptr(y_pooled)), 'rmem_stem7x7s2_pool',
              (x_padded, w4, bias, y_pooled))


def conv3x3_direct(x, w, bias, y, *, H, W, C, images, relu=False) -> Op:
    """y = act(conv3x3(x) + bias), C -> C channels (C = 64 or 128), stride 1, pad 1; x / y [images*H*W, C], w [C, 3, 3, C]"""
    _dev(x, w, bias, y)
    dt = w.dtype
    assert C in (64, 128) and x.dtype == dt and y.dtype == dt and bias.dtype == F32 and w.is_contiguous() and w.numel() == 9 * C * C and bias.numel() == C
    assert x.numel() >= images * H * W * C and y.numel() >= images * H * W * C
    return Op(_fn('rmem_conv3x3_direct', dt), (_ptr(x), images, H, W, C, _ptr(w), _ptr(bias), int(relu), _ptr(y)), 'rmem_conv3x3_direct', (x, w, bias, y))


def ingest_rgb8(rgb, *, Hs, Ws, Hd, Wd, out_chw=None, out_nhwc8=None) -> Op:
    """uint8 RGB [Hs, Ws, 3] device tensor -> resized, normalised fp32 [3, Hd, Wd] and/or bf16 [Hd*Wd, 8]."""
    _dev(rgb, out_chw, out_nhwc8)
    dt = _first16(out_nhwc8)
    assert rgb.dtype == torch.uint8 and rgb.is_contiguous() and rgb.numel() == Hs * Ws * 3
    assert (out_chw is None or out_chw.dtype == F32) and (out_nhwc8 is None or out_nhwc8.dtype == dt)
    return Op(_fn('rmem_ingest_rgb8', dt), (_ptr(rgb), Hs, Ws, Hd, Wd, _ptr(out_chw), _ptr(out_nhwc8)), 'rmem_ingest_rgb8', (rgb, out_chw, out_nhwc8))


def maxpool3x3s2(x, y, *, H, W, C, images=1) -> Op:
    _dev(x, y)
    dt = x.dtype
    assert x.is_contiguous() and y.is_contiguous() and x.numel() >= images * H * W * C
    assert y.numel() >= images * ((H - 1) // 2 + 1) * ((W - 1) // 2 + 1) * C
    return Op(_fn('rmem_maxpool3x3s2_nhwc_images', dt), (_ptr(x), _ptr(y), images, H, W, C), 'rmem_maxpool3x3s2_nhwc', (x, y))


def bilinear(x, y, *, Hi, Wi, Ho, Wo, C, align_corners=True, images=1) -> Op:
    _dev(x, y)
    dt = x.dtype
    assert x.dtype == dt and y.dtype == dt and x.numel() >= images * Hi * Wi * C and y.numel() >= images * Ho * Wo * C
    return Op(_fn('rmem_bilinear_nhwc_images', dt), (_ptr(x), _ptr(y), images, Hi, Wi, Ho, Wo, C, int(align_corners)),
              'rmem_bilinear_nhwc', (x, y))


def logits_post(logits, *, ldl, nc, keep, Hi, Wi, Ho, Wo, align_corners=True, out=None, label_u8=None, label_f32=None, images=1) -> Op:
    """images > 1: logits [images, Hi*Wi, ldl] -> labels [images, Ho, Wo] (labels only)."""
    _dev(logits, out, label_u8, label_f32)
    assert logits.dtype == F32 and (out is None or out.dtype == F32) and (label_u8 is None or label_u8.dtype == torch.uint8)
    assert logits.numel() >= images * Hi * Wi * ldl and (images == 1 or out is None)
    for lab in (label_u8, label_f32):
        assert lab is None or (lab.is_contiguous() and lab.numel() >= images * Ho * Wo)
    args = (_ptr(logits), images, ldl, nc, keep, Hi, Wi, Ho, Wo, int(align_corners), _ptr(out), _ptr(label_u8), _ptr(label_f32))
    return Op(_lib.lib().rmem_logits_post_images, args, 'rmem_logits_post', (logits, out, label_u8, label_f32))


def label_to_onehot16(label, out, *, Hs, Ws, Hd, Wd, ncls=11, images=1) -> Op:
    """label [images, Hs, Ws] uint8 / fp32 -> bf16 [images, Hd*Wd, 16]"""
    _dev(label, out)
    dt = out.dtype
    assert label.dtype in (torch.uint8, F32) and label.is_contiguous() and out.dtype == dt
    assert label.numel() >= images * Hs * Ws and out.numel() >= images * Hd * Wd * 16
    args = (_ptr(label), int(label.dtype == F32), images, Hs, Ws, Hd, Wd, ncls, _ptr(out))
    return Op(_fn('rmem_label_to_onehot16_images', dt), args, 'rmem_label_to_onehot16', (label, out))


def label_id_embed_scratch(images: int, H: int, W: int, pad: int, device) -> torch.Tensor:
    """the bordered uint8 label map rmem_label_id_embed works in: [images, Hpd, Wpd], filled with 255 (the border stays that way)"""
    hp, wp = C.c_int(0), C.c_int(0)
    if _lib.lib().rmem_label_id_embed_scratch_size(H, W, pad, C.byref(hp), C.byref(wp)):
        raise RmemError('rmem_label_id_embed_scratch_size: bad argument')
    return torch.full((images, hp.value, wp.value), 255, dtype=torch.uint8, device=device)


def label_id_embed(label, w, bias, scratch_u8, out, *, Hs, Ws, H, W, K, stride, pad, ncls=11, images=1) -> Op:
    """label [images, Hs, Ws] uint8 / fp32 -> id embedding [images * Ho * Wo, 256] = bias + conv(one_hot(label at H x W), w) without the one-hot
    tensor; w [256, K, K, 16]; scratch_u8 from label_id_embed_scratch(images, H, W, pad)"""
    _dev(label, w, bias, scratch_u8, out)
    dt = w.dtype
    ho, wo = (H + 2 * pad - K) // stride + 1, (W + 2 * pad - K) // stride + 1
    assert label.dtype in (torch.uint8, F32) and label.is_contiguous() and label.numel() >= images * Hs * Ws
    assert w.is_contiguous() and w.numel() == 256 * K * K * 16 and bias.dtype == F32 and bias.numel() == 256 and out.dtype == dt
    assert scratch_u8.dtype == torch.uint8 and scratch_u8.numel() >= images * (H + 2 * pad + 1) * (W + 2 * pad) and out.numel() >= images * ho * wo * 256
    args = (_ptr(label), int(label.dtype == F32), images, Hs, Ws, H, W, K, K, stride, pad, ncls, _ptr(w), _ptr(bias), _ptr(scratch_u8), _ptr(out))
    return Op(_fn('rmem_label_id_embed', dt), args, 'rmem_label_id_embed', (label, w, bias, scratch_u8, out))


def resize_nearest_flip(src, dst, *, flip: bool) -> Op:
    """fp32 [..., Hs, Ws] -> nearest resize to dst's [..., Hd, Wd] (same leading planes), optionally flipped along W."""
    _dev(src, dst)
    assert src.dtype == F32 and dst.dtype == F32 and src.is_contiguous() and dst.is_contiguous()
    Hs, Ws, Hd, Wd = int(src.shape[-2]), int(src.shape[-1]), int(dst.shape[-2]), int(dst.shape[-1])
    planes = src.numel() // (Hs * Ws)
    assert dst.numel() == planes * Hd * Wd
    return Op(_lib.lib().rmem_resize_nearest_flip_f32, (_ptr(src), planes, Hs, Ws, _ptr(dst), Hd, Wd, int(flip)), 'rmem_resize_nearest_flip_f32',
              (src, dst))


def evict_scores(logits, mass, scores, *, ldl, nc, keep, Hi, Wi, He, We, T) -> Op:
    _dev(logits, mass, scores)
    assert scores.dtype == F32 and scores.numel() >= 32 + 64 * 32
    args = (_ptr(logits), ldl, nc, keep, Hi, Wi, He, We, _ptr(mass), T, _ptr(scores))
    return Op(_lib.lib().rmem_evict_scores, args, 'rmem_evict_scores', (logits, mass, scores))


def copy_async(dst, src, nbytes: int) -> Op:
    """dst/src: torch tensors (device, or pinned host); plain byte copy on the launch stream."""
    assert dst.is_contiguous() and src.is_contiguous()
    assert nbytes <= dst.numel() * dst.element_size() and nbytes <= src.numel() * src.element_size()
    return Op(_lib.lib().rmem_copy_async, (_ptr(dst), _ptr(src), nbytes), 'rmem_copy_async', (dst, src))


class PinnedRing:
    """Pinned host staging rows for small tables that are re-sent to the device while earlier frames are still queued
    (bank chunk tables, append-slot tables).  A row is rewritten only after the H2D copy that last read it has EXECUTED:
    every upload records a HIP event on the launch stream and ``next()`` synchronises that row's event (normally long
    signalled, so this is free); graph replay enqueues ~10x faster than the GPU runs, so a plain modulo ring would wrap
    under a queued copy."""

    def __init__(self, rows: int, shape, dtype, device):
        self.host = torch.zeros(rows, *shape, dtype=dtype).pin_memory()
        self.events = [None] * rows
        self.device = device
        self.i = 0

    def next(self) -> torch.Tensor:
        self.i = (self.i + 1) % self.host.shape[0]
        ev = self.events[self.i]
        if ev is not None:
            ev.synchronize()
        return self.host[self.i]

    def upload(self, dst: torch.Tensor, nbytes: int, stream: int):
        """Enqueue host row -> dst on ``stream`` and remember when it has been read."""
        copy_async(dst, self.host[self.i], nbytes)(stream)
        ev = self.events[self.i]
        if ev is None:
            ev = self.events[self.i] = torch.cuda.Event()
        ev.record(torch.cuda.ExternalStream(stream, device=self.device))


def scatter_blocks(src, dst, slots, *, nclips, block_bytes, slot_bytes) -> Op:
    """block c of src -> dst + slots[c] * slot_bytes (slots: device int32 table, negative = skip)."""
    _dev(src, dst, slots)
    assert slots.dtype == torch.int32 and slots.numel() >= nclips and src.numel() * src.element_size() >= nclips * block_bytes
    return Op(_lib.lib().rmem_scatter_blocks, (_ptr(src), _ptr(dst), _ptr(slots), nclips, block_bytes, slot_bytes), 'rmem_scatter_blocks',
              (src, dst, slots))


def copy2d_async(dst, dst_pitch: int, src, src_pitch: int, row_bytes: int, rows: int) -> Op:
    """rows x row_bytes device copy between pitched buffers (pitches in bytes)."""
    _dev(dst, src)
    assert (rows - 1) * dst_pitch + row_bytes <= dst.numel() * dst.element_size()
    assert (rows - 1) * src_pitch + row_bytes <= src.numel() * src.element_size()
    return Op(_lib.lib().rmem_copy2d_async, (_ptr(dst), dst_pitch, _ptr(src), src_pitch, row_bytes, rows), 'rmem_copy2d_async', (dst, src))


def gated_workspace(Lq: int, DV: int, frames: int, keys_per_frame: int, nchunks: int, device, nclips: int = 1) -> torch.Tensor:
    n = _lib.lib().rmem_gated_attn_workspace_bytes(Lq, DV, frames, keys_per_frame, nchunks)
    if n == 0:
        raise RmemError('rmem_gated_attn_workspace_bytes: bad geometry')
    return torch.empty(nclips * n // 4 + 64, dtype=F32, device=device)


def gated_attn(q, k_bank, v_bank, u_a, out, workspace, *, Lq, DV, ldq, ldk, ldv, ldua, ldo, k_slot_stride=0, v_slot_stride=0,
               chunks=None, nchunks=1, frames=1, keys_per_frame, pe_cur=None, pe_mem=None, u_b=None, ldub=0, usplit=None,
               mass=None, dw=None, H=0, W=0, nclips=1) -> Op:
    """DeAOT gated propagation attention (single head, d_att 128); see include/rmem.h.  nclips > 1: [clip][rows][ld] operands."""
    _dev(q, k_bank, v_bank, u_a, u_b, out, workspace, chunks, pe_cur, pe_mem, mass, dw)
    dt = q.dtype
    assert dw is None or (dw.dtype == F32 and dw.numel() == 25 * DV and H * W == Lq)
    assert all(t.dtype == dt for t in (q, k_bank, v_bank, u_a, out)) and (u_b is None or u_b.dtype == dt)
    assert workspace.numel() * 4 >= nclips * _lib.lib().rmem_gated_attn_workspace_bytes(Lq, DV, frames, keys_per_frame, nchunks)
    assert chunks is None or (chunks.dtype == torch.int32 and chunks.numel() >= nclips * nchunks * 8)
    assert mass is None or (mass.dtype == F32 and mass.numel() >= nclips * Lq * frames)
    assert out.numel() >= ((nclips - 1) * Lq + Lq - 1) * ldo + DV
    usplit = DV if usplit is None else usplit
    args = (_ptr(q), ldq, _ptr(k_bank), k_slot_stride, ldk, _ptr(v_bank), v_slot_stride, ldv, _ptr(chunks), nchunks, frames,
            keys_per_frame, _ptr(pe_cur), _ptr(pe_mem), Lq, DV, _ptr(u_a), ldua, _ptr(u_b), ldub, usplit, _ptr(out), ldo,
            _ptr(mass), _ptr(dw), H, W, nclips, _ptr(workspace))
    return Op(_fn('rmem_gated_attn_clips', dt), args, 'rmem_gated_attn',
              (q, k_bank, v_bank, u_a, u_b, out, workspace, chunks, pe_cur, pe_mem, mass, dw))


def local_gated_attn(q, k, v, rel, u_a, out, workspace, *, H, W, DV, ldq, ldk, ldv, ldrel, ldua, ldo, u_b=None, ldub=0,
                     usplit=None, dw=None, nclips=1) -> Op:
    """DeAOT 15x15 local gated propagation attention; rel = relative_emb_k(q) fp32 [H*W][ldrel].  nclips > 1: [clip][rows][ld] operands."""
    _dev(q, k, v, rel, u_a, u_b, out, workspace, dw)
    dt = q.dtype
    assert dw is None or (dw.dtype == F32 and dw.numel() == 25 * DV)
    assert all(t.dtype == dt for t in (q, k, v, u_a, out)) and rel.dtype == F32
    assert workspace.numel() * 4 >= nclips * _lib.lib().rmem_gated_attn_workspace_bytes(H * W, DV, 1, H * W, 8)
    usplit = DV if usplit is None else usplit
    args = (_ptr(q), ldq, _ptr(k), ldk, _ptr(v), ldv, _ptr(rel), ldrel, H, W, DV, _ptr(u_a), ldua, _ptr(u_b), ldub, usplit,
            _ptr(out), ldo, _ptr(dw), nclips, _ptr(workspace))
    return Op(_fn('rmem_local_gated_attn_clips', dt), args, 'rmem_local_gated_attn', (q, k, v, rel, u_a, u_b, out, workspace, dw))


class Graph:
    """A captured launch list (hipGraph) replayable on any stream."""

    def __init__(self, ops, stream: int):
        L = _lib.lib()
        _lib.check(L.rmem_graph_begin(stream), 'rmem_graph_begin')
        try:
            for o in ops:
                o(stream)
        finally:
            h = C.c_void_p()
            rc = L.rmem_graph_end(stream, C.byref(h))
        _lib.check(rc, 'rmem_graph_end')
        self.handle = h
        self.keep = ops

    def __call__(self, stream: int):
        _lib.check(_lib.lib().rmem_graph_launch(self.handle, stream), 'rmem_graph_launch')

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().rmem_graph_destroy(self.handle)
        except Exception:
            pass
