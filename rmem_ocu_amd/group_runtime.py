"""Device state and launch lists for a GROUP of clips that advance in lockstep (R50-AOTL and SwinB-AOTL paths).

At HW = 1674 tokens a launch costs about as much as its arithmetic and only four kernels are in flight on the GPU
(DESIGN.md §7b), so the throughput path does not run one clip per launch list: B clips of equal length share one
``GroupRuntime``.  Every activation buffer of runtime.ClipRuntime gets a leading clip dimension ([B * rows, C], clip-major),
which turns

    every linear / LayerNorm / add      into the same launch over B * HW rows,
    every convolution                   into one launch over a batch of B images (rmem_conv_desc.batch),
    GroupNorm / depth-wise / bilinear   into the per-image batched forms (statistics per clip),
    the three attentions                into one launch with a clip dimension (rmem_mem_read_attn_clips).

The clips are independent; what differs between them is the memory bank's content and, after evictions, its slot order:
the bank is [B * slots, HW, 256] per layer, clip c owns slots c * S .. c * S + S - 1, the chunk table has one block of rows
per clip, and appends go through a device table of destination slots (rmem_scatter_blocks), so ONE captured hipGraph per
bank size serves the whole group.  Frame counters, append schedule and bank size T are the same for all clips of a group
(equal length => same gap, evaluator.py:330-335); the eviction decision is per clip (engines/group_engine.py).

The per-op arithmetic is exactly ClipRuntime's (same kernels, same operands per clip), cf. the citations there.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import os

import torch

from . import ops
from .encoder_batch import BatchEncoder, SwinBatchEncoder
from .pack import R50_BLOCKS, R50_STRIDES
from .runtime import BF16, D_MODEL, F32, FFN, HEADS, MAX_CHUNKS, PLAIN_CHUNKS, _out, sine_pos_emb, temporal_slots


class GroupRuntime:
    deaot = False
    max_rows = MAX_CHUNKS                 # key-table rows per clip
    bank_kw = bank_vw = D_MODEL           # widths of a bank entry's key / value rows
    dec_cin = 4 * D_MODEL                 # columns of the decoder's LSTT input

    def __init__(self, P: Dict[str, torch.Tensor], in_hw: Tuple[int, int], bank_slots: int, device, clips: int,
                 num_lstt: int = 3, align_corners: bool = True, num_classes: int = 11, lookahead: int = 4):
        if ('g0.qvu.w' in P) != self.deaot:
            raise ops.RmemError('GroupRuntime covers the AOTL paths (ResNet-50 / Swin-B), group_runtime_deaot.GroupRuntimeDeAOT the R50-DeAOTL path')
        self.swin = 'pe.w' in P
        self.chain = not __import__('os').environ.get('RMEM_NO_CHAIN')      # 1: the unfused launch list (A/B runs, identity tests)
        self.pair_attn = not __import__('os').environ.get('RMEM_NO_PAIR_ATTN')   # 1: memory read and short-term attention as two launches
        self.P, self.dev, self.NL, self.B = P, device, num_lstt, clips
        self.dt = P['proj.w'].dtype
        self.align, self.nc = align_corners, num_classes
        H, W = in_hw
        self.H, self.W = H, W
        B = clips
        if self.swin:     # Swin-B (cfg 5): patch 4 and two patch mergings (runtime.ClipRuntime has the citations)
            if H % 4 or W % 4:
                raise ops.RmemError('Swin-B path: network size must be a multiple of 4 (the evaluator makes it a multiple of 16)')
            self.H4, self.W4 = H // 4, W // 4
            self.H8, self.W8 = (self.H4 + 1) // 2, (self.W4 + 1) // 2
            self.H16, self.W16 = (self.H8 + 1) // 2, (self.W8 + 1) // 2
            self.enc_ch = (128, 256, 512)
        else:
            self.H2, self.W2 = _out(H, 7, 2, 3), _out(W, 7, 2, 3)
            self.H4, self.W4 = _out(self.H2, 3, 2, 1), _out(self.W2, 3, 2, 1)
            self.H8, self.W8 = _out(self.H4, 3, 2, 1), _out(self.W4, 3, 2, 1)
            self.H16, self.W16 = _out(self.H8, 3, 2, 1), _out(self.W8, 3, 2, 1)
            self.enc_ch = (256, 512, 1024)
        self.L = self.H16 * self.W16
        L, M4, M8 = self.L, self.H4 * self.W4, self.H8 * self.W8
        self.M4, self.M8 = M4, M8
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or self.dt, device=device)  # noqa: E731
        # ---- encoders: one for the frame in flight (reference frames), one running `lookahead` frames ahead; image = e * B + c
        Enc = SwinBatchEncoder if self.swin else BatchEncoder
        self.enc_now = Enc(P, in_hw, B, device)
        self.lookahead = lookahead
        # two look-ahead encoders: while the frames of one batch are being propagated, the next batch is encoded on the engine's
        # side stream (group_engine.py); look-ahead slot s = buffer * lookahead + frame
        self.enc_bufs = [Enc(P, in_hw, B * lookahead, device) for _ in range(2)] if lookahead > 1 else []
        self.enc_ahead = self.enc_bufs[0] if self.enc_bufs else None
        # ---- LSTT buffers, [B * L, .] clip-major ----
        R = B * L
        self.dec_in = e(R, self.dec_cin)
        self.id_emb = e(R, D_MODEL)
        self.onehot = e(B * H * W, 16)
        self._alloc_lstt(R, num_lstt)
        self.gn_ws = ops.groupnorm_workspace(32, device, images=B)
        # GroupNorm partial sums of the FFN hidden written by the chained linear1 (one entry per 32-row block, the rest stays zero)
        self.ffn_stats = torch.zeros(B * 32 * 64 * 2, dtype=F32, device=device)
        self.chain_stats = self.chain and (self.L + 31) // 32 <= 64 and not __import__('os').environ.get('RMEM_NO_CHAIN_STATS')
        self.conv_ws = torch.empty(16 * R * D_MODEL, dtype=F32, device=device)
        self.mass = torch.zeros(B * L * MAX_CHUNKS, dtype=F32, device=device)          # [B][L][T] compact for the current T
        self.scores = torch.zeros(B, 32 + 64 * 32, dtype=F32, device=device)
        self.scores_host = torch.zeros(B, MAX_CHUNKS, dtype=F32).pin_memory()
        # ---- decoder buffers ----
        self.d16a, self.d16b = e(R, 256), e(R, 256)
        self.d8a, self.d8b = e(B * M8, 256), e(B * M8, 256)
        self.d4a, self.d4b = e(B * M4, 128), e(B * M4, 128)
        self.logits = torch.zeros(B * M4, 16, dtype=F32, device=device)
        # ---- memory bank: [B * S, L, width] per layer, clip c owns slots c * S .. ----
        self.S = bank_slots
        self.bank_K = [e(B * bank_slots, L, self.bank_kw) for _ in range(num_lstt)]
        self.bank_V = [e(B * bank_slots, L, self.bank_vw) for _ in range(num_lstt)]
        self.slots: List[List[int]] = [[] for _ in range(B)]            # per clip: logical order t -> local slot
        self.free: List[List[int]] = [list(range(bank_slots)) for _ in range(B)]
        self.chunks = torch.zeros(B * self.max_rows, 8, dtype=torch.int32, device=device)
        self.chunks_ring = ops.PinnedRing(4, (B * self.max_rows, 8), torch.int32, device)
        self.append_slots = torch.full((B,), -1, dtype=torch.int32, device=device)   # global destination slot per clip
        self.append_ring = ops.PinnedRing(4, (B,), torch.int32, device)
        self._prog: Dict[str, list] = {}

    def _alloc_lstt(self, R: int, num_lstt: int):
        """LSTT activations, [B * L, .] clip-major."""
        device, L, B = self.dev, self.L, self.B
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or self.dt, device=device)  # noqa: E731
        self.x = e(R, D_MODEL, dt=F32)
        self.t1b = e(R, D_MODEL)
        self.qkv = e(R, 3 * D_MODEL)
        self.att = e(R, D_MODEL)
        self.att2 = e(R, D_MODEL)           # short-term attention output (the chained route keeps both attentions' outputs)
        self.t3 = e(R, D_MODEL)
        self.k4, self.v4 = e(R, D_MODEL), e(R, D_MODEL)
        self.h1, self.h3 = e(R, FFN), e(R, FFN)
        self.curr_Q = [e(R, D_MODEL) for _ in range(num_lstt)]
        self.curr_V = [e(R, D_MODEL) for _ in range(num_lstt)]
        self.new_V = [e(R, D_MODEL) for _ in range(num_lstt)]       # linear_V(curr_V + id): the bank entry before it is scattered
        self.tgt3 = [e(R, D_MODEL) for _ in range(num_lstt)]
        self.short_K = [e(R, D_MODEL) for _ in range(num_lstt)]
        self.short_V = [e(R, D_MODEL) for _ in range(num_lstt)]
        self.tmpA = [e(R, D_MODEL) for _ in range(num_lstt)]
        self.tmpB = [e(R, D_MODEL) for _ in range(num_lstt)]
        pos = sine_pos_emb(self.H16, self.W16).to(device)
        self.posb = pos.to(self.dt).repeat(B, 1).contiguous()
        self.pos_qk = [torch.zeros(R, 3 * D_MODEL, dtype=F32, device=device) for _ in range(num_lstt)]
        self._pos_ready = False
        self.attn_ws = ops.attn_workspace(L, HEADS, MAX_CHUNKS, device, nclips=B)

    # ------------------------------------------------------------------ bank bookkeeping (host) + device tables
    def reset_bank(self):
        self.slots = [[] for _ in range(self.B)]
        self.free = [list(range(self.S)) for _ in range(self.B)]

    @property
    def T(self) -> int:
        """Bank size the group's launches are built for: the LONGEST bank among the clips (after a mid-clip reference frame a
        clip's bank restarts at one entry, aot_engine.py:322; shorter banks are padded with empty table rows)."""
        return max(len(s) for s in self.slots)

    def chunk_plan(self, T: int) -> Tuple[int, int]:
        if T > MAX_CHUNKS:
            raise ops.RmemError(f'memory bank of {T} frames exceeds the {MAX_CHUNKS}-chunk table')
        splits = max(1, min(8 // T, MAX_CHUNKS // T))
        return splits, T * splits

    def _keys_per_chunk(self, splits: int) -> int:
        return (self.L + splits - 1) // splits

    def _chunk_rows(self, slots: List[List[int]]):
        """Chunk-table rows (global slot, key begin, key count, temporal-PE slot, t), one block of n rows per clip, for per-clip slot
        orders.  Banks may differ in length: the block is laid out for the longest one (same key split for every clip), a shorter
        bank uses its own temporal-PE slots (layers/transformer.py:598-621 depends on that clip's T) and its block is padded with
        empty rows (key_count 0), which the kernel reads as zero keys with zero mass."""
        T = max(len(s) for s in slots)
        splits, n = self.chunk_plan(T)
        per = self._keys_per_chunk(splits)
        rows = []
        for c in range(self.B):
            pes = temporal_slots(len(slots[c]))
            mine = []
            for t, s in enumerate(slots[c]):
                for kb in range(0, self.L, per):
                    mine.append((c * self.S + s, kb, min(per, self.L - kb), pes[t], t))
            mine += [(c * self.S, 0, 0, -1, 0)] * (n - len(mine))
            rows += mine
        return rows, n

    def upload_chunks(self, stream: int):
        """One block of rows per clip for the current (equal) bank size; bank slots are global indexes c * S + local slot."""
        rows, n = self._chunk_rows(self.slots)
        host = self.chunks_ring.next()          # waits for the upload that last read this row (normally long done)
        host.zero_()
        host[:len(rows), :5] = torch.tensor(rows, dtype=torch.int32)
        self.chunks_ring.upload(self.chunks, self.B * n * 8 * 4, stream)

    def upload_append_slots(self, local_slots: List[int], stream: int):
        host = self.append_ring.next()
        for c, s in enumerate(local_slots):
            host[c] = c * self.S + s if s >= 0 else -1
        self.append_ring.upload(self.append_slots, self.B * 4, stream)

    def mem_read_probe(self, T: int, layer: int = 0):
        """The long-term memory read of ``layer`` at bank size T as a stand-alone Op over this runtime's own buffers (bench.py's
        roofline leg): same shapes, chunk plan and kernel as prog_lstt's (there the launch also carries the short-term attention's
        workgroups, rmem_lstt_attn_pair_clips: not here, so that the timed launch does exactly the FLOPs returned), but with its own chunk
        table (bank slots 0 .. T-1 of every clip in logical order) so the engine's device state is untouched.  Returns (op, algorithmic FLOPs)."""
        if not 1 <= T <= self.S:
            raise ops.RmemError(f'mem_read_probe: T = {T} outside 1..{self.S}')
        rows, n = self._chunk_rows([list(range(T)) for _ in range(self.B)])
        table = torch.zeros(self.B * n, 8, dtype=torch.int32)
        table[:, :5] = torch.tensor(rows, dtype=torch.int32)
        self._probe_chunks = table.to(self.dev)
        L, C, i = self.L, D_MODEL, layer
        op = self._attn(self.curr_Q[i], C, self.bank_K[i], self.bank_V[i], C, self.att, slot_stride=L * C, chunks=self._probe_chunks,
                        nchunks=n, lk_single=T * L, pe_cur=self.P['pe_cur'], pe_mem=self.P['pe_mem'], mass=None, T=T)
        return op, 4.0 * L * (T * L) * C * self.B

    # ------------------------------------------------------------------ helpers
    def _conv(self, *a, **kw):
        return ops.conv2d(*a, ws=self.conv_ws, batch=self.B, **kw)

    def _lin(self, x, name, y, K, N, **kw):
        return ops.linear(x, self.P[name + '.w'], self.P[name + '.b'], y, M=self.B * self.L, K=K, N=N, ws=self.conv_ws, **kw)

    def _attn(self, q, ldq, k, v, ldkv, out, **kw):
        L = self.L
        return ops.mem_read_attn(q, k, v, out, self.attn_ws, Lq=L, heads=HEADS, ldq=ldq, ldkv=ldkv, ldo=D_MODEL, nclips=self.B,
                                 q_cs=L * ldq, out_cs=L * D_MODEL, **kw)

    def _scatter(self, src, bank):
        nb = self.L * D_MODEL * 2
        return ops.scatter_blocks(src, bank, self.append_slots, nclips=self.B, block_bytes=nb, slot_bytes=nb)

    def prepare_pos(self, stream: int):
        if self._pos_ready:
            return
        for i in range(self.NL):
            ops.run(ops.linear(self.posb, self.P[f'l{i}.self_qk.w'], None, self.pos_qk[i], M=self.B * self.L, K=D_MODEL, N=2 * D_MODEL,
                               ldo=3 * D_MODEL), stream)
        self._pos_ready = True

    def _enc(self, e: Optional[int]):
        """(enc1, enc2, enc3) [B * rows, C] of the frame in flight: the group's own encoder or look-ahead slot e."""
        if e is None:
            return self.enc_now.enc_out
        B = self.B
        buf, f = divmod(e, self.lookahead)
        o = self.enc_bufs[buf].enc_out
        return tuple(t[f * B:(f + 1) * B] for t in o)

    # ------------------------------------------------------------------ programs
    def prog_encode(self) -> list:
        return self.enc_now.prog()

    def prog_project(self, e: Optional[int]) -> list:
        key = f'project_{e}'
        if key not in self._prog:
            self._prog[key] = [ops.conv2d(self._enc(e)[2], self.P['proj.w'], self.P['proj.b'], self.x, H=self.B * self.L, W=1,
                                          Cin=self.enc_ch[2], Cout=D_MODEL, y2=self.dec_in, ld2=4 * D_MODEL, ws=self.conv_ws)]
        return self._prog[key]

    def prog_lstt(self, ref_mode: bool, T: int, want_mass: bool = True) -> list:
        """ClipRuntime.prog_lstt over B clips (layers/transformer.py:553-692).  ref_mode: the frame's own K / V are computed
        into curr_Q / new_V and scattered into each clip's first bank slot before the long-term read."""
        key = 'lstt_ref' if ref_mode else f'lstt_prop{T}{"m" if want_mass else ""}'
        if key in self._prog:
            return self._prog[key]
        P, L, B, o = self.P, self.L, self.B, []
        C = D_MODEL
        R = B * L
        _, nchunks = self.chunk_plan(1 if ref_mode else T)
        if not ref_mode and self.chain:
            self._prog[key] = self._prog_lstt_chained(T, nchunks, want_mass)
            return self._prog[key]
        for i in range(self.NL):
            d = f'l{i}'
            o.append(ops.layernorm256(self.x, P[d + '.ln1.g'], P[d + '.ln1.b'], M=R, y=self.t1b))
            o.append(self._lin(self.t1b, d + '.self_qkv', self.qkv, C, 3 * C, residual=self.pos_qk[i]))
            o.append(self._attn(self.qkv, 3 * C, self.qkv.view(-1)[C:], self.qkv.view(-1)[2 * C:], 3 * C, self.att,
                                nchunks=PLAIN_CHUNKS, lk_single=L, kv_cs=L * 3 * C))
            o.append(self._lin(self.att, d + '.self_proj', self.x, C, C, residual=self.x))
            o.append(ops.layernorm256(self.x, P[d + '.ln2.g'], P[d + '.ln2.b'], M=R, y=self.curr_V[i]))
            cq = self.curr_Q[i]
            o.append(self._lin(self.curr_V[i], d + '.linear_Q', cq, C, C))
            if ref_mode:
                o.append(ops.add16(self.curr_V[i], self.id_emb, self.tmpB[i], R * C))
                o.append(self._lin(self.tmpB[i], d + '.linear_V', self.new_V[i], C, C))
                o.append(self._scatter(cq, self.bank_K[i]))
                o.append(self._scatter(self.new_V[i], self.bank_V[i]))
                sk, sv = cq, self.new_V[i]                          # local_K/V = global_K/V (transformer.py:585-586)
            else:
                sk, sv = self.short_K[i], self.short_V[i]
            o.append(self._attn(cq, C, self.bank_K[i], self.bank_V[i], C, self.att, slot_stride=L * C, chunks=self.chunks,
                                nchunks=nchunks, lk_single=(1 if ref_mode else T) * L, pe_cur=P['pe_cur'], pe_mem=P['pe_mem'],
                                mass=self.mass if (i == 0 and not ref_mode and want_mass) else None, T=T))
            o.append(self._lin(self.att, d + '.long_proj', self.x, C, C, residual=self.x))
            o.append(ops.layernorm256_pair(sk, cq, self.k4, sv, self.curr_V[i], self.v4, P[d + '.ln4.g'], P[d + '.ln4.b'], M=R))
            o.append(self._attn(cq, C, self.k4, self.v4, C, self.att, nchunks=PLAIN_CHUNKS, lk_single=L, kv_cs=L * C))
            o.append(self._lin(self.att, d + '.short_proj', self.x, C, C, residual=self.x, y2=self.tgt3[i]))
            if ref_mode:   # short-term memory of the reference frame (675-678)
                o.append(self._lin(self.tgt3[i], d + '.linear_QMem', self.short_K[i], C, C))
                o.append(ops.add16(self.tgt3[i], self.id_emb, self.tmpA[i], R * C))
                o.append(self._lin(self.tmpA[i], d + '.linear_VMem', self.short_V[i], C, C))
            o.append(ops.layernorm256(self.x, P[d + '.ln3.g'], P[d + '.ln3.b'], M=R, y=self.t3))
            o.append(self._lin(self.t3, d + '.linear1', self.h1, C, FFN))
            o.append(ops.gn_act_dwconv5x5(self.h1, P[d + '.gn.g'], P[d + '.gn.b'], P[d + '.dw.w'], self.h3, self.gn_ws, H=self.H16,
                                          W=self.W16, C=FFN, groups=32, act=2, images=B))
            o.append(self._lin(self.h3, d + '.linear2', self.x, FFN, C, residual=self.x))
            o.append(ops.layernorm256(self.x, P[f'dec_norm{i}.g'], P[f'dec_norm{i}.b'], M=R,
                                      y=self.dec_in.view(-1)[(i + 1) * C:], ldy=4 * C))
        self._prog[key] = o
        return o

    def _prog_lstt_chained(self, T: int, nchunks: int, want_mass: bool) -> list:
        """Propagate-mode LSTT with the row-local sequences between the attentions as ONE launch each (csrc/rowchain.hip):
        7 launches per block instead of 16, bit-identical buffers (x, curr_Q, curr_V, k4, v4, tgt3, h1, qkv, dec_in) to the
        unfused list of prog_lstt.  The long-term projection moves behind the short-term attention (which does not read x)."""
        P, L, B, C, o = self.P, self.L, self.B, D_MODEL, []
        ln = lambda n: (P[n + '.g'], P[n + '.b'])       # noqa: E731

        def chain_c(i_done, i_next):
            kw = {}
            if i_done is not None:
                d = f'l{i_done}'
                kw.update(h3=self.h3, w2=P[d + '.linear2.wf'], b2=P[d + '.linear2.b'], dec_norm=ln(f'dec_norm{i_done}'),
                          dec_out=self.dec_in.view(-1)[(i_done + 1) * C:], ld_dec=self.dec_cin)
            if i_next is not None:
                d = f'l{i_next}'
                kw.update(ln1=ln(d + '.ln1'), w_qkv=P[d + '.self_qkv.wf'], b_qkv=P[d + '.self_qkv.b'], pos_qk=self.pos_qk[i_next], qkv=self.qkv)
            return ops.lstt_chain_c(L=L, clips=B, x=self.x, dt=self.dt, **kw)

        o.append(chain_c(None, 0))
        for i in range(self.NL):
            d = f'l{i}'
            cq = self.curr_Q[i]
            o.append(self._attn(self.qkv, 3 * C, self.qkv.view(-1)[C:], self.qkv.view(-1)[2 * C:], 3 * C, self.att,
                                nchunks=PLAIN_CHUNKS, lk_single=L, kv_cs=L * 3 * C))
            o.append(ops.lstt_chain_a(L=L, clips=B, att=self.att, x=self.x, w_proj=P[d + '.self_proj.wf'], b_proj=P[d + '.self_proj.b'],
                                      ln2=ln(d + '.ln2'), curr_v=self.curr_V[i], w_q=P[d + '.linear_Q.wf'], b_q=P[d + '.linear_Q.b'], curr_q=cq,
                                      short_k=self.short_K[i], short_v=self.short_V[i], ln4=ln(d + '.ln4'), k4=self.k4, v4=self.v4))
            mass = self.mass if (i == 0 and want_mass) else None
            if self.pair_attn:       # long-term memory read + short-term attention: one launch (same queries, independent keys)
                o.append(ops.lstt_attn_pair(cq, self.bank_K[i], self.bank_V[i], self.att, self.k4, self.v4, self.att2, self.attn_ws, Lq=L,
                                            heads=HEADS, ldq=C, ldkv=C, ldo=C, slot_stride=L * C, chunks=self.chunks, nchunks=nchunks,
                                            lk_total=T * L, pe_cur=P['pe_cur'], pe_mem=P['pe_mem'], mass=mass, T=T, nclips=B, q_cs=L * C,
                                            out_cs=L * C, lk_short=L, kv_short_cs=L * C, out_short_cs=L * C))
            else:
                o.append(self._attn(cq, C, self.bank_K[i], self.bank_V[i], C, self.att, slot_stride=L * C, chunks=self.chunks,
                                    nchunks=nchunks, lk_single=T * L, pe_cur=P['pe_cur'], pe_mem=P['pe_mem'], mass=mass, T=T))
                o.append(self._attn(cq, C, self.k4, self.v4, C, self.att2, nchunks=PLAIN_CHUNKS, lk_single=L, kv_cs=L * C))
            o.append(ops.lstt_chain_b(L=L, clips=B, att_long=self.att, att_short=self.att2, x=self.x, w_long=P[d + '.long_proj.wf'],
                                      b_long=P[d + '.long_proj.b'], w_short=P[d + '.short_proj.wf'], b_short=P[d + '.short_proj.b'],
                                      tgt3=self.tgt3[i], ln3=ln(d + '.ln3'), w1=P[d + '.linear1.wf'], b1=P[d + '.linear1.b'], h1=self.h1,
                                      gn_partial=self.ffn_stats if self.chain_stats else None, gn_splits=64 if self.chain_stats else 0))
            if self.chain_stats:     # the GroupNorm statistics of the FFN hidden came out of linear1's epilogue (<= 64 row blocks per clip)
                o.append(ops.gn_act_dwconv5x5_prestats(self.h1, P[d + '.gn.g'], P[d + '.gn.b'], P[d + '.dw.w'], self.h3, self.ffn_stats,
                                                       H=self.H16, W=self.W16, C=FFN, groups=32, act=2, images=B))
            else:
                o.append(ops.gn_act_dwconv5x5(self.h1, P[d + '.gn.g'], P[d + '.gn.b'], P[d + '.dw.w'], self.h3, self.gn_ws, H=self.H16,
                                              W=self.W16, C=FFN, groups=32, act=2, images=B))
            o.append(chain_c(i, i + 1 if i + 1 < self.NL else None))
        return o

    def prog_decode(self, e: Optional[int]) -> list:
        key = f'decode_{e}'
        if key in self._prog:
            return self._prog[key]
        P, o, L, B = self.P, [], self.L, self.B
        M8, M4 = self.M8, self.M4
        enc1, enc2, enc3 = self._enc(e)
        gn = lambda x, name, y, M, C: ops.groupnorm(x, P[name + '.gn.g'], P[name + '.gn.b'], y, self.gn_ws, M=M, C=C, groups=8, act=1,  # noqa: E731
                                                    images=B)
        lin = lambda x, name, y, M, K, N, **kw: ops.conv2d(x, P[name + '.w'], P[name + '.b'], y, H=B * M, W=1, Cin=K, Cout=N,  # noqa: E731
                                                           ws=self.conv_ws, **kw)
        o.append(lin(self.dec_in, 'dec.conv_in', self.d16a, L, self.dec_cin, 256))
        o.append(gn(self.d16a, 'dec.conv_in', self.d16b, L, 256))
        c4, c8, c16 = self.enc_ch
        o.append(lin(enc3, 'dec.adapter_16x', self.d16a, L, c16, 256, residual=self.d16b))
        o.append(self._conv(self.d16a, P['dec.conv_16x.w'], P['dec.conv_16x.b'], self.d16b, H=self.H16, W=self.W16, Cin=256, Cout=256,
                            KH=3, KW=3, pad=1))
        o.append(gn(self.d16b, 'dec.conv_16x', self.d16a, L, 256))
        import os
        fuse_up = not os.environ.get('RMEM_NO_UPFUSE')      # timing experiments only
        # F.interpolate(x, size) + adapter(shortcut) (decoders/fpn.py:49-52): the resize happens in the GEMM's residual read
        if fuse_up:
            o.append(ops.conv2d(enc2, P['dec.adapter_8x.w'], P['dec.adapter_8x.b'], self.d8b, H=self.H8, W=self.W8, Cin=c8, Cout=256,
                                batch=B, residual=self.d16a, res_up=(self.H16, self.W16, self.align)))
        else:
            o.append(ops.bilinear(self.d16a, self.d8a, Hi=self.H16, Wi=self.W16, Ho=self.H8, Wo=self.W8, C=256, align_corners=self.align,
                                  images=B))
            o.append(lin(enc2, 'dec.adapter_8x', self.d8b, M8, c8, 256, residual=self.d8a))
        d8c = self.d8a.view(-1)[: B * M8 * 128]
        o.append(self._conv(self.d8b, P['dec.conv_8x.w'], P['dec.conv_8x.b'], d8c, H=self.H8, W=self.W8, Cin=256, Cout=128, KH=3, KW=3,
                            pad=1))
        d8d = self.d8b.view(-1)[: B * M8 * 128]
        o.append(gn(d8c, 'dec.conv_8x', d8d, M8, 128))
        if fuse_up:
            o.append(ops.conv2d(enc1, P['dec.adapter_4x.w'], P['dec.adapter_4x.b'], self.d4b, H=self.H4, W=self.W4, Cin=c4, Cout=128,
                                batch=B, residual=d8d, res_up=(self.H8, self.W8, self.align)))
        else:
            o.append(ops.bilinear(d8d, self.d4a, Hi=self.H8, Wi=self.W8, Ho=self.H4, Wo=self.W4, C=128, align_corners=self.align, images=B))
            o.append(lin(enc1, 'dec.adapter_4x', self.d4b, M4, c4, 128, residual=self.d4a))
        if os.environ.get('RMEM_DIRECT_CONV3_128') == '1':   # rows kept in LDS, weights in registers (bit-identical; 97 -> 75 us alone, neutral in the pipeline: opt-in)
            o.append(ops.conv3x3_direct(self.d4b, P['dec.conv_4x.w'], P['dec.conv_4x.b'], self.d4a, H=self.H4, W=self.W4, C=128, images=B))
        else:
            o.append(self._conv(self.d4b, P['dec.conv_4x.w'], P['dec.conv_4x.b'], self.d4a, H=self.H4, W=self.W4, Cin=128, Cout=128, KH=3,
                                KW=3, pad=1))
        if os.environ.get('RMEM_NO_HEADFUSE'):               # timing experiments only
            o.append(gn(self.d4a, 'dec.conv_4x', self.d4b, M4, 128))
            o.append(lin(self.d4b, 'dec.conv_out', self.logits, M4, 128, self.nc, ldo=16))
        else:                                                # conv_out(relu(gn(x))) in one pass over x (decoders/fpn.py:62-66)
            o.append(ops.groupnorm_head(self.d4a, P['dec.conv_4x.gn.g'], P['dec.conv_4x.gn.b'], P['dec.conv_out.w'], P['dec.conv_out.b'],
                                        self.logits, self.gn_ws, M=M4, C=128, groups=8, N=self.nc, ldy=16, act=1, images=B))
        self._prog[key] = o
        return o

    def prog_id_emb(self, labels: torch.Tensor, hs: int, ws: int) -> list:
        """labels: uint8 or fp32 [B, hs, ws] device tensor at a fixed address -> self.id_emb (aot_engine.py:208-232)."""
        key = f'id_{labels.data_ptr()}_{hs}_{ws}'
        if key in self._prog:
            return self._prog[key]
        P, B = self.P, self.B
        k, s, p = (17, 16, 8) if self.align else (16, 16, 0)
        if os.environ.get('RMEM_NO_LABEL_EMBED', '0') != '1':
            # the one-hot operand of the id bank's conv is built in registers from the label bytes (csrc/idbank.hip): no one-hot tensor
            if getattr(self, 'lab_net', None) is None:
                self.lab_net = ops.label_id_embed_scratch(B, self.H, self.W, p, self.dev)
            o = [ops.label_id_embed(labels, P['idbank.w'], P['idbank.b'], self.lab_net, self.id_emb, Hs=hs, Ws=ws, H=self.H, W=self.W, K=k,
                                    stride=s, pad=p, ncls=self.nc, images=B)]
        else:
            o = [ops.label_to_onehot16(labels, self.onehot, Hs=hs, Ws=ws, Hd=self.H, Wd=self.W, ncls=self.nc, images=B)]
            o.append(self._conv(self.onehot, P['idbank.w'], P['idbank.b'], self.id_emb, H=self.H, W=self.W, Cin=16, Cout=D_MODEL,
                                KH=k, KW=k, stride=s, pad=p))
        self._prog[key] = o
        return o

    def prog_update(self, append: bool) -> list:
        """Memory update of the frame just propagated for all clips (layers/transformer.py:269-322); with ``append`` the new bank
        entries are scattered to the per-clip slots named by the device table."""
        key = f'update_{int(append)}'
        if key in self._prog:
            return self._prog[key]
        P, NL, L, C, B = self.P, self.NL, self.L, D_MODEL, self.B
        R = B * L
        o = [ops.add16_grouped(self.tgt3 + (self.curr_V if append else []), [self.id_emb] * (NL * (2 if append else 1)),
                                  self.tmpA + (self.tmpB if append else []), R * C)]
        w = lambda nm: [P[f'l{i}.{nm}.w'] for i in range(NL)]   # noqa: E731
        b = lambda nm: [P[f'l{i}.{nm}.b'] for i in range(NL)]   # noqa: E731
        o.append(ops.linear_grouped(self.tgt3, w('linear_QMem'), b('linear_QMem'), self.short_K, M=R, K=C, N=C))
        o.append(ops.linear_grouped(self.tmpA, w('linear_VMem'), b('linear_VMem'), self.short_V, M=R, K=C, N=C))
        if append:
            o.append(ops.linear_grouped(self.tmpB, w('linear_V'), b('linear_V'), self.new_V, M=R, K=C, N=C))
            for i in range(NL):
                o.append(self._scatter(self.curr_Q[i], self.bank_K[i]))
                o.append(self._scatter(self.new_V[i], self.bank_V[i]))
        self._prog[key] = o
        return o
