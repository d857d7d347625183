"""CPU oracle of the R50-DeAOTL(+RMem) inference path -- TEST INFRASTRUCTURE ONLY.

fp32 torch-functional restatement of the reference's DeAOT variant (the model the shipped eval_vost.sh runs): every
function cites the reference file:line it follows.  Pinned by tests/golden/deaot_*.npz, which were produced by running the
reference itself (tests/golden/make_golden.py deaot).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product path (rmem_ocu_amd) never does.

What differs from the AOT path (oracle/ref_cpu.py):
  * the propagation block is GatedPropagationModule (layers/transformer.py:1011-1249): single-head gated attention with
    d_att = 128, values [V | ID_V] 1024 wide, gate U, depth-wise 5x5 and a 1024 -> 512 projection (layers/attention.py:93-216),
    a 15x15 local window attention with a learned relative embedding (attention.py:220-413), and a second (ID) residual
    stream;
  * the stack is DualBranchGPM (transformer.py:700-1008): final GroupNorm1D(512, 2 groups) only, the decoder sees the last
    layer (models/deaot.py:28-40, configs/models/default_deaot.py:13), the identity embedding is LayerNorm'ed
    (deaot.py:64-68), the temporal embedding is 128 wide (deaot.py:46-53);
  * restrict_long_memories has no "bank not full yet" early return (transformer.py:880-892 vs 331-333): the EMA scores and
    visit counts are updated on EVERY long-term update, a drop happens only when the bank overflows.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from .ref_cpu import (OracleEngine, W, assign_identity, choose_eviction, fpn_decode, ln, temporal_slots)

Tensor = torch.Tensor
D_ATT = 128
MAX_DIS = 7
WIN = 2 * MAX_DIS + 1


def silu(x: Tensor) -> Tensor:
    """attention.py:89-90."""
    return x * torch.sigmoid(x)


def dw_conv(x: Tensor, size_2d: Tuple[int, int], weight: Tensor) -> Tensor:
    """basic.py:38-57 (eval: dropout off): depth-wise 5x5, no bias, on [HW,B,C]."""
    h, wd = size_2d
    _, bs, c = x.shape
    x = x.view(h, wd, bs, c).permute(2, 3, 0, 1)
    x = F.conv2d(x, weight, None, padding=2, groups=c)
    return x.reshape(bs, c, h * wd).permute(2, 0, 1)


def gated_propagation(Q: Tensor, K: Tensor, V: Tensor, U: Tensor, size_2d, w: W, p: str, use_linear: bool,
                      explicit: bool) -> Tuple[Tensor, Optional[Tensor]]:
    """attention.py:138-216 with num_head = 1.  Q [Lq,B,.], K [Lk,B,.], V [Lk,B,.], U [Lq,B,.] -> ([Lq,B,512], attn)."""
    l, bs, _ = Q.shape
    if use_linear:                                               # self-attention flavour (151-173)
        Q = K = F.linear(Q, w[p + '.linear_QK.weight'], w[p + '.linear_QK.bias'])
        half = V.shape[-1] // 2
        V = silu(torch.cat([F.linear(V[..., :half], w[p + '.linear_V1.weight'], w[p + '.linear_V1.bias']),
                            F.linear(V[..., half:], w[p + '.linear_V2.weight'], w[p + '.linear_V2.bias'])], dim=-1))
        U = silu(torch.cat([F.linear(U[..., :half], w[p + '.linear_U1.weight'], w[p + '.linear_U1.bias']),
                            F.linear(U[..., half:], w[p + '.linear_U2.weight'], w[p + '.linear_U2.bias'])], dim=-1))
    hid = V.shape[-1]
    if explicit:                                                 # 175-195
        q = (Q / D_ATT ** 0.5).view(-1, bs, 1, D_ATT).permute(1, 2, 0, 3)
        k = K.view(-1, bs, 1, D_ATT).permute(1, 2, 3, 0)
        v = V.view(-1, bs, 1, hid).permute(1, 2, 0, 3)
        attn = torch.softmax(q @ k, dim=-1)
        out = (attn @ v).permute(2, 0, 1, 3)
    else:                                                        # 197-206
        q = Q.view(-1, bs, 1, D_ATT).permute(1, 2, 0, 3)
        k = K.view(-1, bs, 1, D_ATT).permute(1, 2, 0, 3)
        v = V.view(-1, bs, 1, hid).permute(1, 2, 0, 3)
        out = F.scaled_dot_product_attention(q, k, v, None, 0.0, is_causal=False).permute(2, 0, 1, 3)
        attn = None
    out = out.reshape(l, bs, -1) * U                             # 208
    out = dw_conv(out, size_2d, w[p + '.dw_conv.conv.weight'])   # 210
    return F.linear(out, w[p + '.projection.weight'], w[p + '.projection.bias']), attn


def _pad_unfold(x: Tensor) -> Tensor:
    """attention.py:403-413: zero pad by 7, unfold 15x15 windows -> [N, C*225, HW]."""
    x = F.pad(x, (MAX_DIS, MAX_DIS, MAX_DIS, MAX_DIS), mode='constant', value=0)
    return F.unfold(x, kernel_size=(WIN, WIN), stride=(1, 1), dilation=1)


def local_gated_propagation(q: Tensor, k: Tensor, v: Tensor, u: Tensor, size_2d, w: W, p: str) -> Tensor:
    """attention.py:281-363 (use_linear False, num_head 1, enable_corr False, fp32).
    q, k [1,128,h,w]; v [1,1024,h,w]; u [HW,1,1024] -> [HW,1,512]."""
    n, c, h, wd = v.shape
    ones = torch.ones((1, 1, h, wd))
    qk_mask = 1 - _pad_unfold(ones).view(1, 1, WIN * WIN, h * wd)                       # 299-303
    rel = F.conv2d(q, w[p + '.relative_emb_k.weight'], w[p + '.relative_emb_k.bias'])   # 305 (un-scaled q)
    rel = rel.view(n, 1, WIN * WIN, h * wd)
    qs = (q / D_ATT ** 0.5).view(-1, D_ATT, h, wd)                                      # 308-310
    unf_k = _pad_unfold(k.view(-1, D_ATT, h, wd)).view(n, D_ATT, WIN * WIN, h, wd)      # 323-326
    qk = (qs.unsqueeze(2) * unf_k).sum(dim=1).view(n, 1, WIN * WIN, h * wd)             # 327-328
    qk = qk + rel
    qk = qk - qk_mask * 1e+8                                                            # 338
    attn = torch.softmax(qk, dim=2)                                                     # 340
    # local2global + matmul (344-347, 365-401): out-of-map taps carry exactly zero probability, so the scatter into the
    # padded map followed by the crop equals a window-gathered weighted sum
    unf_v = _pad_unfold(v).view(n, c, WIN * WIN, h * wd)
    agg = (unf_v * attn.view(n, 1, WIN * WIN, h * wd)).sum(dim=2)                       # [n, c, HW]
    agg = agg.permute(2, 0, 1)
    out = agg * u                                                                       # 349
    out = dw_conv(out, size_2d, w[p + '.dw_conv.conv.weight'])
    return F.linear(out, w[p + '.projection.weight'], w[p + '.projection.bias'])


def seq_to_2d(t: Tensor, size_2d) -> Tensor:
    """basic.py:73-77."""
    h, wd = size_2d
    _, n, c = t.shape
    return t.view(h, wd, n, c).permute(2, 3, 0, 1).contiguous()


def fuse_id(value: Optional[Tensor], id_emb: Tensor, w: W, p: str) -> Tensor:
    """transformer.py:1236-1242 (the key half is always None)."""
    x = id_emb if value is None else torch.cat([value, id_emb], dim=2)
    return silu(F.linear(x, w[p + '.linear_ID_V.weight'], w[p + '.linear_ID_V.bias']))


def gpm_block(tgt: Tensor, tgt_id: Optional[Tensor], w: W, p: str, long_mem, short_mem, curr_id_emb, size_2d,
              temporal: Optional[Tensor], save_attn: bool):
    """transformer.py:1091-1234.  Returns (tgt, tgt_id, memories, record)."""
    d_model = tgt.shape[-1]
    _tgt = ln(tgt, w, p + '.norm1')
    qv = F.linear(_tgt, w[p + '.linear_QV.weight'], w[p + '.linear_QV.bias'])
    curr_Q = curr_K = qv[..., :D_ATT]
    local_Q = seq_to_2d(curr_Q, size_2d)
    curr_V = silu(qv[..., D_ATT:])
    curr_U = F.linear(_tgt, w[p + '.linear_U.weight'], w[p + '.linear_U.bias'])
    if tgt_id is None:                                          # 1115-1119
        tgt_id = 0
        cat_U = torch.cat([silu(curr_U), torch.ones_like(curr_U)], dim=-1)
        curr_ID_V = None
    else:                                                       # 1120-1124
        _tgt_id = ln(tgt_id, w, p + '.id_norm1')
        curr_ID_V = _tgt_id
        cat_U = silu(torch.cat([curr_U, F.linear(_tgt_id, w[p + '.linear_ID_U.weight'], w[p + '.linear_ID_U.bias'])], dim=-1))

    if curr_id_emb is not None:                                 # 1126-1136
        global_K, global_V = curr_K, curr_V
        local_K, local_V = seq_to_2d(global_K, size_2d), seq_to_2d(global_V, size_2d)
        global_ID_V = fuse_id(curr_ID_V, curr_id_emb, w, p)
        local_ID_V = seq_to_2d(global_ID_V, size_2d)
        global_K, global_V, global_ID_V = global_K[None], global_V[None], global_ID_V[None]
    else:
        global_K, global_V, _, global_ID_V = long_mem
        local_K, local_V, _, local_ID_V = short_mem

    T, L, bs, E = global_K.shape                                # 1141-1177
    if temporal is not None:
        cur_pe, mem_pe = temporal[0:1], temporal[1:]
        pe = mem_pe[temporal_slots(T, mem_pe.shape[0])]
        flat_K = (global_K + pe.view(T, 1, 1, E)).flatten(0, 1)
        q_time = curr_Q + cur_pe.view(1, 1, E)
    else:
        flat_K, q_time = global_K.flatten(0, 1), curr_Q
    cat_global_V = torch.cat([global_V.flatten(0, 1), global_ID_V.flatten(0, 1)], dim=-1)
    cat_local_V = torch.cat([local_V, local_ID_V], dim=1)

    cat_tgt2, attn = gated_propagation(q_time, flat_K, cat_global_V, cat_U, size_2d, w, p + '.long_term_attn', False, save_attn)
    record = None
    if save_attn:                                               # 1185-1192
        record = attn.view(bs, 1, L, T, L).mean(dim=1)[0].sum(dim=2)
    cat_tgt3 = local_gated_propagation(local_Q, local_K, cat_local_V, cat_U, size_2d, w, p + '.short_term_attn')

    tgt = tgt + cat_tgt2[..., :d_model] + cat_tgt3[..., :d_model]          # 1212-1220
    tgt_id = tgt_id + cat_tgt2[..., d_model:] + cat_tgt3[..., d_model:]

    x = torch.cat([ln(tgt, w, p + '.norm2'), ln(tgt_id, w, p + '.id_norm2')], dim=-1)   # 1223-1227
    cat_s, _ = gated_propagation(x, x, x, x, size_2d, w, p + '.self_attn', True, False)
    tgt = tgt + cat_s[..., :d_model]
    tgt_id = tgt_id + cat_s[..., d_model:]
    return tgt, tgt_id, [[curr_K, curr_V, None, curr_ID_V], [global_K, global_V, None, global_ID_V],
                         [local_K, local_V, None, local_ID_V]], record


class OracleDeAOTEngine(OracleEngine):
    """DeAOTEngine (engines/deaot_engine.py:9-19 = AOTEngine on a DeAOT model) driven like DeAOTInferEngine drives it."""

    def __init__(self, weights: W, former_len: int = 1, latter_len: int = 8, long_term_mem_gap: int = 5, num_lstt: int = 3,
                 align_corners: bool = True):
        super().__init__(weights, former_len, latter_len, long_term_mem_gap, num_lstt, align_corners)

    def _assign_identity(self, oh, ign):
        """models/deaot.py:64-68: LayerNorm over the channel of the identity embedding."""
        e = assign_identity(oh, ign, self.w, self.align_corners)          # [HW, B, C]
        return ln(e, self.w, 'id_norm')

    def _lstt(self, xs, id_emb, save_attn):
        """transformer.py:766-823."""
        x = xs[-1].flatten(2).permute(2, 0, 1).contiguous()
        xid, mems, rec0 = None, [], None
        for i in range(self.L):
            x, xid, m, rec = gpm_block(x, xid, self.w, f'LSTT.layers.{i}',
                                       self.long_mem[i] if self.long_mem is not None else None,
                                       self.short_mem[i] if self.short_mem is not None else None,
                                       id_emb, self.enc_size_2d, self.temporal, save_attn)
            mems.append(m)
            if i == 0:
                rec0 = rec
        cat = torch.cat([x, xid], dim=2)
        cat = F.group_norm(cat.permute(1, 2, 0), 2, self.w['LSTT.decoder_norms.0.gn.weight'],
                           self.w['LSTT.decoder_norms.0.gn.bias'], 1e-5).permute(2, 0, 1)      # basic.py:6-12
        self.curr_mem = [m[0] for m in mems]
        self.lstt_long = [m[1] for m in mems]
        self.lstt_short = [m[2] for m in mems]
        if save_attn:
            self.record_attn_weight = rec0
        return [cat]

    def _decode(self, xs, lstt_outs, output_size):
        """models/deaot.py:56-62 + decoders/fpn.py:38-41 (decode_intermediate_input False: the last input only)."""
        n, _, h, wd = xs[-1].shape
        last = lstt_outs[-1].view(h, wd, n, -1).permute(2, 3, 0, 1)
        logits = fpn_decode([last], xs, self.w, self.align_corners)
        self.pred_id_logits = logits
        if output_size is not None:
            logits = F.interpolate(logits, size=output_size, mode='bilinear', align_corners=self.align_corners)
        return logits

    def _update_memories(self, id_emb, update_long):
        """transformer.py:825-872 (update_short_memories / update_long_term_memory) + 874-996 (restriction)."""
        short = []
        for i in range(self.L):
            k, v, _, idv = self.curr_mem[i]
            idv = fuse_id(idv, id_emb, self.w, f'LSTT.layers.{i}')
            self.curr_mem[i][3] = idv
            short.append([seq_to_2d(k, self.enc_size_2d), seq_to_2d(v, self.enc_size_2d), None, seq_to_2d(idv, self.enc_size_2d)])
        self.short_mem = short
        if not update_long:
            return
        for i in range(self.L):
            self.long_mem[i] = [None if (old is None or new is None) else torch.cat([old, new[None]], dim=0)
                                for old, new in zip(self.long_mem[i], self.curr_mem[i])]
        self.long_memories_indexes.append(self.frame_step)
        logits = F.interpolate(self.pred_id_logits, size=self.enc_size_2d, mode='bilinear', align_corners=True)
        fg = 1 - torch.softmax(logits, dim=1)[:, 0:1]
        # no early return here: scores and visit counts move on every long-term update (transformer.py:880-968)
        drop = choose_eviction(self.record_attn_weight, fg.flatten(), self.long_memories_indexes, self.evict)
        if self.long_mem[0][0].shape[0] <= self.former + self.latter:
            return
        self.drop_trace.append(drop)
        for i in range(self.L):
            self.long_mem[i] = [None if m is None else torch.cat([m[:drop], m[drop + 1:]], dim=0) for m in self.long_mem[i]]
        del self.long_memories_indexes[drop]
