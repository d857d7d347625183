"""Per-clip device state and launch lists of the R50-DeAOTL path (the model eval_vost.sh:11 runs).

``DeAOTRuntime`` keeps ClipRuntime's encoder, FPN decoder, identity-bank and memory-ring machinery and replaces the
propagation stack by DualBranchGPM (layers/transformer.py:700-1008; one GatedPropagationModule per layer, 1011-1249):

    per layer   LN(tgt) -> one GEMM [Q | SiLU V | SiLU U] (linear_QV + linear_U)             transformer.py:1102-1111
                layers >= 1: LN(tgt_id) (= curr_ID_V), SiLU(linear_ID_U)                      1120-1124
                long-term gated attention over the bank (K 128 wide, [V | ID_V] 1024 wide)    1141-1184
                15x15 local gated attention over the previous frame                            1199-1200
                each followed by depth-wise 5x5 + projection into BOTH residual streams        attention.py:208-211
                LN(tgt), LN(tgt_id) -> one block-diagonal GEMM -> gated self-attention         1222-1232
    final       GroupNorm1D(512, 2) over [tgt | tgt_id] -> the decoder's only LSTT input       760-808, models/deaot.py:56-62

The two residual streams live side by side in one fp32 [HW, 512] buffer, so every projection (1024 -> 512) adds into both
with one GEMM.  The bank stores what the reference concatenates at read time: K [slots, HW, 128] and [V | ID_V]
[slots, HW, 1024] (transformer.py:1179); the 15x15 window memory is the same pair for the previous frame.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops
from .runtime import BF16, D_MODEL, F32, ClipRuntime

D_ATT, E1, E2 = 128, 512, 1024          # d_att, expand_d_model, expand_d_vu (transformer.py:1027-1034, attention.py:106-117)
QVU = D_ATT + 2 * E1                     # [Q | V | U] columns of the fused GEMM
SQVU = D_ATT + 2 * E2                    # [QK | V | U] columns of the fused self-attention GEMM
REL_LD = 256                             # 225 relative-embedding logits padded to a 16-byte friendly row
GP_ROWS = 64                             # chunk-table capacity of rmem_gated_attn


class DeAOTRuntime(ClipRuntime):
    max_chunks = GP_ROWS
    bank_kw, bank_vw = D_ATT, E2

    # ------------------------------------------------------------------ buffers
    def _alloc_lstt(self, L: int, num_lstt: int):
        dev = self.dev
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or self.dt, device=dev)  # noqa: E731
        self.xc0 = torch.zeros(L, 2 * D_MODEL, dtype=F32, device=dev)   # [encoder projection | 0]: tgt and "tgt_id = 0" of layer 0
        self.xc = e(L, 2 * D_MODEL, dt=F32)                             # [tgt | tgt_id]
        self.x = self.xc0                                               # where the encoder projector writes (ldo 512)
        self.dec_in = e(L, 2 * D_MODEL)
        self.dec_cin = 2 * D_MODEL
        self.n1 = e(L, D_MODEL)
        self.qvu = [e(L, QVU) for _ in range(num_lstt)]                 # Q = curr_K and SiLU(V) = curr_V stay until the update
        self.idcat = [None] + [e(L, 2 * D_MODEL) for _ in range(1, num_lstt)]   # [LN(tgt_id) = curr_ID_V | id_emb]
        self.idu = e(L, E1)
        self.g1, self.g2 = e(L, E2), e(L, E2)
        self.rel = torch.zeros(L, REL_LD, dtype=F32, device=dev)
        self.xn = e(L, 2 * D_MODEL)
        self.sqvu = e(L, SQVU)
        self.short_K = [e(L, D_ATT) for _ in range(num_lstt)]
        self.short_V = [e(L, E2) for _ in range(num_lstt)]
        self.id_emb = e(L, D_MODEL)
        self.id_raw = e(L, D_MODEL, dt=F32)

    def _on_bank_resized(self):
        # probabilities [HW, slots * roundup(HW, 64)] bf16 + split-K slabs, sized for a full ring
        self.gp_ws = ops.gated_workspace(self.L, E2, self.S, self.L, GP_ROWS, self.dev)

    def _proj_op(self, cin: int, enc3=None):
        return self._conv(self.enc3 if enc3 is None else enc3, self.P['proj.w'], self.P['proj.b'], self.xc0, H=self.L, W=1, Cin=cin,
                          Cout=D_MODEL, ldo=2 * D_MODEL)

    def prepare_pos(self, stream: int):
        """GatedPropagationModule never adds the spatial positional embedding (with_pos_embed is unused, 1084-1089)."""

    # ------------------------------------------------------------------ chunk table
    def chunk_plan(self, T: int) -> Tuple[int, int]:
        """About 32 table rows (row boundaries cost nothing in the P.V kernel, and the score kernels get one workgroup per
        (query tile, row)); key ranges start on 64-key tile boundaries."""
        if T > 32:
            raise ops.RmemError(f'memory bank of {T} frames exceeds the 32 frames the gated attention records mass for')
        splits = max(1, min(4, 32 // T))
        per = self._keys_per_chunk(splits)
        return splits, T * ((self.L + per - 1) // per)

    def _keys_per_chunk(self, splits: int) -> int:
        return ((self.L + splits - 1) // splits + 63) // 64 * 64

    def mem_read_probe(self, T: int, layer: int = 0):
        """Stand-alone Op of layer's long-term gated attention at bank size T over this runtime's own buffers (bench.py's roofline
        leg): same launch as prog_lstt's, own chunk table (bank slots 0 .. T-1), no mass, so engine state is untouched."""
        if not 1 <= T <= self.S:
            raise ops.RmemError(f'mem_read_probe: T = {T} outside 1..{self.S}')
        rows, n = self._chunk_rows(list(range(T)))
        table = torch.zeros(n, 8, dtype=torch.int32)
        table[:, :5] = torch.tensor(rows, dtype=torch.int32)
        self._probe_chunks = table.to(self.dev)
        i, L, P = layer, self.L, self.P
        ua, ldua, ub = self._gate(i)
        op = ops.gated_attn(self.qvu[i], self.bank_K[i], self.bank_V[i], ua, self.g2, self.gp_ws, Lq=L, DV=E2, ldq=QVU,
                            ldk=D_ATT, ldv=E2, ldua=ldua, ldo=E2, k_slot_stride=L * D_ATT, v_slot_stride=L * E2,
                            chunks=self._probe_chunks, nchunks=n, frames=T, keys_per_frame=L, pe_cur=P['pe_cur'],
                            pe_mem=P['pe_mem'], u_b=ub, ldub=E1, usplit=E1, mass=None,
                            dw=P[f'g{i}.long_dw.w'], H=self.H16, W=self.W16)
        return op, 2.0 * L * (T * L) * (D_ATT + E2)

    # ------------------------------------------------------------------ programs
    def _gate(self, i: int):
        """(u_a, ldua, u_b): cat_curr_U = [SiLU(U) | ones] in layer 0, [SiLU(U) | SiLU(ID_U)] after (1115-1124)."""
        return self.qvu[i].view(-1)[D_ATT + E1:], QVU, (self.idu if i > 0 else None)

    def _tail(self, name: str, residual, **kw):
        """projection (attention.py:211) added into both residual streams; the depth-wise 5x5 before it (210) runs inside
        the attention call's combine launch and leaves its result in g2."""
        return [self._lin(self.g2, name + '_proj', self.xc, self.L, E2, 2 * D_MODEL, residual=residual, **kw)]

    def prog_lstt(self, ref_mode: bool, T: int, ref_slot: int = 0, want_mass: bool = True) -> list:
        key = f'lstt_ref{ref_slot}' if ref_mode else f'lstt_prop{T}{"m" if want_mass else ""}'
        if key in self._prog:
            return self._prog[key]
        P, L, o = self.P, self.L, []
        C = D_MODEL
        _, nchunks = self.chunk_plan(1 if ref_mode else T)
        frames = 1 if ref_mode else T
        for i in range(self.NL):
            d = f'g{i}'
            xin = self.xc0 if i == 0 else self.xc
            o.append(ops.layernorm256(xin, P[d + '.ln1.g'], P[d + '.ln1.b'], M=L, lda=2 * C, y=self.n1))
            o.append(self._lin(self.n1, d + '.qvu', self.qvu[i], L, C, QVU, relu=3, act_begin=D_ATT))
            if i > 0:
                o.append(ops.layernorm256(self.xc.view(-1)[C:], P[d + '.idn1.g'], P[d + '.idn1.b'], M=L, lda=2 * C,
                                          y=self.idcat[i], ldy=2 * C))
                o.append(self._lin(self.idcat[i], d + '.idu', self.idu, L, C, E1, relu=3, ldx=2 * C))
            if ref_mode:
                # the frame is its own memory (1126-1136): K, V and ID_V = SiLU(linear_ID_V([curr_ID_V | id_emb])) go
                # straight into bank slot ref_slot and into the window memory
                o += self._write_memory(i, self.bank_K[i][ref_slot], self.bank_V[i][ref_slot])
                o += self._copy_memory(i, self.bank_K[i][ref_slot], self.bank_V[i][ref_slot], self.short_K[i], self.short_V[i])
            ua, ldua, ub = self._gate(i)
            o.append(ops.gated_attn(self.qvu[i], self.bank_K[i], self.bank_V[i], ua, self.g2, self.gp_ws, Lq=L, DV=E2, ldq=QVU,
                                    ldk=D_ATT, ldv=E2, ldua=ldua, ldo=E2, k_slot_stride=L * D_ATT, v_slot_stride=L * E2,
                                    chunks=self.chunks, nchunks=nchunks, frames=frames, keys_per_frame=L, pe_cur=P['pe_cur'],
                                    pe_mem=P['pe_mem'], u_b=ub, ldub=E1, usplit=E1,
                                    mass=self.mass if (i == 0 and not ref_mode and want_mass) else None,
                                    dw=P[d + '.long_dw.w'], H=self.H16, W=self.W16))
            o += self._tail(d + '.long', xin)
            o.append(self._lin(self.qvu[i], d + '.rel', self.rel, L, D_ATT, 225, ldo=REL_LD, ldx=QVU))
            o.append(ops.local_gated_attn(self.qvu[i], self.short_K[i], self.short_V[i], self.rel, ua, self.g2, self.gp_ws,
                                          H=self.H16, W=self.W16, DV=E2, ldq=QVU, ldk=D_ATT, ldv=E2, ldrel=REL_LD, ldua=ldua, ldo=E2,
                                          u_b=ub, ldub=E1, usplit=E1, dw=P[d + '.short_dw.w']))
            o += self._tail(d + '.short', self.xc)
            # --- gated self-attention over [LN(tgt) | LN(tgt_id)] (1222-1232)
            o.append(ops.layernorm256(self.xc, P[d + '.ln2.g'], P[d + '.ln2.b'], M=L, lda=2 * C, y=self.xn, ldy=2 * C))
            o.append(ops.layernorm256(self.xc.view(-1)[C:], P[d + '.idn2.g'], P[d + '.idn2.b'], M=L, lda=2 * C,
                                      y=self.xn.view(-1)[C:], ldy=2 * C))
            o.append(self._lin(self.xn, d + '.self', self.sqvu, L, 2 * C, SQVU, relu=3, act_begin=D_ATT))
            o.append(ops.gated_attn(self.sqvu, self.sqvu, self.sqvu.view(-1)[D_ATT:], self.sqvu.view(-1)[D_ATT + E2:], self.g2,
                                    self.gp_ws, Lq=L, DV=E2, ldq=SQVU, ldk=SQVU, ldv=SQVU, ldua=SQVU, ldo=E2, nchunks=8, frames=1,
                                    keys_per_frame=L, dw=P[d + '.self_dw.w'], H=self.H16, W=self.W16))
            o += self._tail(d + '.self', self.xc)
        o.append(ops.groupnorm(self.xc, P['dec_gn.g'], P['dec_gn.b'], self.dec_in, self.gn_ws, M=L, C=2 * C, groups=2))
        self._prog[key] = o
        return o

    def _write_memory(self, i: int, k_dst: torch.Tensor, v_dst: torch.Tensor) -> list:
        """Memory entry of the current frame for layer i: K = curr_K, V = curr_V (copies out of the fused GEMM's output),
        ID_V = SiLU(linear_ID_V([curr_ID_V | id_emb])) (fuse_key_value_id, 1236-1242) written beside V."""
        L, C = self.L, D_MODEL
        o = [ops.copy2d_async(k_dst, D_ATT * 2, self.qvu[i], QVU * 2, D_ATT * 2, L),
             ops.copy2d_async(v_dst, E2 * 2, self.qvu[i].view(-1)[D_ATT:], QVU * 2, E1 * 2, L)]
        if i == 0:
            o.append(self._lin(self.id_emb, 'g0.idv', v_dst.view(-1)[E1:], L, C, E1, relu=3, ldo=E2))
        else:
            o.append(ops.copy2d_async(self.idcat[i].view(-1)[C:], 2 * C * 2, self.id_emb, C * 2, C * 2, L))
            o.append(self._lin(self.idcat[i], f'g{i}.idv', v_dst.view(-1)[E1:], L, 2 * C, E1, relu=3, ldo=E2))
        return o

    def _copy_memory(self, i: int, k_src, v_src, k_dst, v_dst) -> list:
        L = self.L
        return [ops.copy_async(k_dst, k_src, L * D_ATT * 2), ops.copy_async(v_dst, v_src, L * E2 * 2)]

    def prog_id_emb(self, label: torch.Tensor, hs: int, ws: int) -> list:
        """label map -> one-hot -> identity bank conv -> LayerNorm (models/deaot.py:64-68)."""
        key = f'id_{label.data_ptr()}_{hs}_{ws}'
        if key in self._prog:
            return self._prog[key]
        P = self.P
        k, s, p = (17, 16, 8) if self.align else (16, 16, 0)
        o = [ops.label_to_onehot16(label, self.onehot, Hs=hs, Ws=ws, Hd=self.H, Wd=self.W, ncls=self.nc),
             self._conv(self.onehot, P['idbank.w'], P['idbank.b'], self.id_raw, H=self.H, W=self.W, Cin=16, Cout=D_MODEL,
                        KH=k, KW=k, stride=s, pad=p),
             ops.layernorm256(self.id_raw, P['idnorm.g'], P['idnorm.b'], M=self.L, y=self.id_emb)]
        self._prog[key] = o
        return o

    def prog_update(self, append_slot: Optional[int]) -> list:
        """update_short_memories / update_long_term_memory (transformer.py:825-872): the window memory of every layer
        becomes the frame just propagated; with append_slot the same entry joins the bank."""
        key = f'update_{append_slot}'
        if key in self._prog:
            return self._prog[key]
        o = []
        for i in range(self.NL):
            o += self._write_memory(i, self.short_K[i], self.short_V[i])
            if append_slot is not None:
                o += self._copy_memory(i, self.short_K[i], self.short_V[i], self.bank_K[i][append_slot], self.bank_V[i][append_slot])
        self._prog[key] = o
        return o
