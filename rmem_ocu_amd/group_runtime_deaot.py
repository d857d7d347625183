"""Device state and launch lists for a GROUP of clips on the R50-DeAOTL path (throughput mode of runtime_deaot.DeAOTRuntime).

Same idea as group_runtime.GroupRuntime: B clips of equal length advance in lockstep and share one set of launch lists, so a
frame of the whole group costs the host one hipGraph launch.  What is batched and what is not:

    every linear / LayerNorm / copy        one launch over B * HW rows (weights are shared, rows are independent);
    convolutions, GroupNorm of the decoder  one launch over a batch of B images (statistics per clip);
    the three gated attentions of a layer   ONE LAUNCH SEQUENCE PER CLIP (rmem_gated_attn / rmem_local_gated_attn have no clip
                                            dimension: their split-K slabs and probability matrix are a per-call workspace) --
                                            they run back to back on the group's stream over the clip's rows of the shared
                                            buffers, against the clip's slots of the shared bank and its block of the key table;
    the final GroupNorm1D over fp32 rows    per clip (statistics per clip, fp32 input form of the kernel).

Bank layout: K [B * S, HW, 128] and [V | ID_V] [B * S, HW, 1024] per layer, clip c owns slots c * S ..; appends go through the
device table of destination slots (rmem_scatter_blocks) exactly as in GroupRuntime, so one captured graph per bank size serves
every frame.  All clips of a group hold banks of the same length (no mid-clip reference frames on this path: the engine refuses).

The per-op arithmetic is DeAOTRuntime's (layers/transformer.py:1011-1249; citations there).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch

from . import ops
from .group_runtime import GroupRuntime
from .runtime import D_MODEL, F32
from .runtime_deaot import D_ATT, E1, E2, GP_ROWS, QVU, REL_LD, SQVU


def rows_per_frame(L: int, T: int, clips: int) -> int:
    """Key-table rows per memory frame of a group's long-term gated attention.  A row is one workgroup per query tile and clip in the
    score kernels, and every workgroup pays a fixed prologue (Q fragments, the temporal-PE bias through the matrix pipe, the row
    maxima) worth ~3 key tiles -- so a frame is cut only as far as filling the GPU needs (~512 workgroups), never into more than 4 rows
    or more than 32 rows in all (the table the mass output describes); with 8 clips per launch and T >= 5 a frame stays one row."""
    wgs = ((L + 127) // 128) * T * clips
    return max(1, min(4, 32 // T, (512 + wgs - 1) // wgs))


class GroupRuntimeDeAOT(GroupRuntime):
    deaot = True
    max_rows = GP_ROWS
    bank_kw, bank_vw = D_ATT, E2
    dec_cin = 2 * D_MODEL

    # ------------------------------------------------------------------ buffers
    def _alloc_lstt(self, R: int, num_lstt: int):
        dev, L = self.dev, self.L
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or self.dt, device=dev)  # noqa: E731
        self.xc0 = torch.zeros(R, 2 * D_MODEL, dtype=F32, device=dev)   # [encoder projection | 0]
        self.xc = e(R, 2 * D_MODEL, dt=F32)                             # [tgt | tgt_id]
        self.n1 = e(R, D_MODEL)
        self.qvu = [e(R, QVU) for _ in range(num_lstt)]
        self.idcat = [None] + [e(R, 2 * D_MODEL) for _ in range(1, num_lstt)]
        self.idu = e(R, E1)
        self.g2 = e(R, E2)
        self.rel = torch.zeros(R, REL_LD, dtype=F32, device=dev)
        self.xn = e(R, 2 * D_MODEL)
        self.sqvu = e(R, SQVU)
        self.short_K = [e(R, D_ATT) for _ in range(num_lstt)]
        self.short_V = [e(R, E2) for _ in range(num_lstt)]
        self.id_raw = e(R, D_MODEL, dt=F32)
        self.gp_ws = None               # sized with the bank (needs self.S): see _gp()

    def _gp(self) -> torch.Tensor:
        if self.gp_ws is None:
            self.gp_ws = ops.gated_workspace(self.L, E2, self.S, self.L, GP_ROWS, self.dev, nclips=self.B)
        return self.gp_ws

    def prepare_pos(self, stream: int):
        """GatedPropagationModule never adds the spatial positional embedding (with_pos_embed is unused, 1084-1089)."""

    # ------------------------------------------------------------------ key table (same plan as DeAOTRuntime.chunk_plan)
    def chunk_plan(self, T: int) -> Tuple[int, int]:
        if T > 32:
            raise ops.RmemError(f'memory bank of {T} frames exceeds the 32 frames the gated attention records mass for')
        splits = rows_per_frame(self.L, T, self.B)
        per = self._keys_per_chunk(splits)
        return splits, T * ((self.L + per - 1) // per)

    def _keys_per_chunk(self, splits: int) -> int:
        return ((self.L + splits - 1) // splits + 63) // 64 * 64

    # ------------------------------------------------------------------ helpers
    def _clip(self, t: torch.Tensor, c: int, ld: int, col: int = 0) -> torch.Tensor:
        """Clip c's rows of a [B * L, ld] buffer, starting at column col (flat view: the ops take a pointer and a row stride)."""
        return t.view(-1)[c * self.L * ld + col:]

    def _long_attn(self, i: int, chunks: torch.Tensor, nchunks: int, frames: int, mass: Optional[torch.Tensor]):
        """The long-term gated attention of layer i for ALL clips of the group: one launch per kernel (clip dimension)."""
        P, L = self.P, self.L
        ub = self.idu if i > 0 else None
        return ops.gated_attn(self.qvu[i], self.bank_K[i], self.bank_V[i], self.qvu[i].view(-1)[D_ATT + E1:],
                              self.g2, self._gp(), Lq=L, DV=E2, ldq=QVU, ldk=D_ATT, ldv=E2, ldua=QVU, ldo=E2,
                              k_slot_stride=L * D_ATT, v_slot_stride=L * E2, chunks=chunks, nchunks=nchunks, frames=frames,
                              keys_per_frame=L, pe_cur=P['pe_cur'], pe_mem=P['pe_mem'], u_b=ub, ldub=E1, usplit=E1, mass=mass,
                              dw=P[f'g{i}.long_dw.w'], H=self.H16, W=self.W16, nclips=self.B)

    def _tail(self, name: str, residual) -> list:
        """depth-wise 5x5 ran inside the attention's combine launch (per clip, into g2); the projection into both residual streams
        (attention.py:211) is one GEMM over the group."""
        return [self._lin(self.g2, name + '_proj', self.xc, E2, 2 * D_MODEL, residual=residual)]

    def mem_read_probe(self, T: int, layer: int = 0):
        """The group's long-term gated attention of ``layer`` at bank size T as a stand-alone Op (bench.py's roofline leg): same launch
        as prog_lstt's (all clips), own key table (bank slots 0 .. T-1), no mass.  Returns (op, algorithmic FLOPs of that one call)."""
        if not 1 <= T <= self.S:
            raise ops.RmemError(f'mem_read_probe: T = {T} outside 1..{self.S}')
        rows, n = GroupRuntime._chunk_rows(self, [list(range(T)) for _ in range(self.B)])
        table = torch.zeros(self.B * n, 8, dtype=torch.int32)
        table[:, :5] = torch.tensor(rows, dtype=torch.int32)
        self._probe_chunks = table.to(self.dev)
        return self._long_attn(layer, self._probe_chunks, n, T, None), 2.0 * self.L * (T * self.L) * (D_ATT + E2) * self.B

    # ------------------------------------------------------------------ programs
    def prog_project(self, e: Optional[int]) -> list:
        key = f'project_{e}'
        if key not in self._prog:
            self._prog[key] = [ops.conv2d(self._enc(e)[2], self.P['proj.w'], self.P['proj.b'], self.xc0, H=self.B * self.L, W=1,
                                          Cin=self.enc_ch[2], Cout=D_MODEL, ldo=2 * D_MODEL, ws=self.conv_ws)]
        return self._prog[key]

    def prog_lstt(self, ref_mode: bool, T: int, want_mass: bool = True) -> list:
        """DeAOTRuntime.prog_lstt over B clips.  ref_mode: every clip's frame is its own memory (1126-1136): the entry is written
        to the window memory and scattered to the clip's first bank slot before the long-term read."""
        key = 'lstt_ref' if ref_mode else f'lstt_prop{T}{"m" if want_mass else ""}'
        if key in self._prog:
            return self._prog[key]
        P, L, B, o = self.P, self.L, self.B, []
        C = D_MODEL
        R = B * L
        frames = 1 if ref_mode else T
        _, nchunks = self.chunk_plan(frames)
        # key ranges of the one-frame self-attention (score workgroups per query tile and clip): as few as still fill the GPU
        tiles_q = ((L + 127) // 128) * B
        self_rows = int(os.environ.get('RMEM_GP_SELF_ROWS', max(2, min(8, (448 + tiles_q - 1) // tiles_q))))
        for i in range(self.NL):
            d = f'g{i}'
            xin = self.xc0 if i == 0 else self.xc
            o.append(ops.layernorm256(xin, P[d + '.ln1.g'], P[d + '.ln1.b'], M=R, lda=2 * C, y=self.n1))
            o.append(self._lin(self.n1, d + '.qvu', self.qvu[i], C, QVU, relu=3, act_begin=D_ATT))
            if i > 0:
                o.append(ops.layernorm256(self.xc.view(-1)[C:], P[d + '.idn1.g'], P[d + '.idn1.b'], M=R, lda=2 * C, y=self.idcat[i],
                                          ldy=2 * C))
                o.append(self._lin(self.idcat[i], d + '.idu', self.idu, C, E1, relu=3, ldx=2 * C))
            if ref_mode:
                o += self._write_memory(i) + self._scatter_memory(i)
            o.append(self._long_attn(i, self.chunks, nchunks, frames, self.mass if (i == 0 and not ref_mode and want_mass) else None))
            o += self._tail(d + '.long', xin)
            o.append(self._lin(self.qvu[i], d + '.rel', self.rel, D_ATT, 225, ldo=REL_LD, ldx=QVU))
            o.append(ops.local_gated_attn(self.qvu[i], self.short_K[i], self.short_V[i], self.rel, self.qvu[i].view(-1)[D_ATT + E1:], self.g2,
                                          self._gp(), H=self.H16, W=self.W16, DV=E2, ldq=QVU, ldk=D_ATT, ldv=E2, ldrel=REL_LD, ldua=QVU,
                                          ldo=E2, u_b=self.idu if i > 0 else None, ldub=E1, usplit=E1, dw=P[d + '.short_dw.w'], nclips=B))
            o += self._tail(d + '.short', self.xc)
            # --- gated self-attention over [LN(tgt) | LN(tgt_id)] (1222-1232)
            o.append(ops.layernorm256(self.xc, P[d + '.ln2.g'], P[d + '.ln2.b'], M=R, lda=2 * C, y=self.xn, ldy=2 * C))
            o.append(ops.layernorm256(self.xc.view(-1)[C:], P[d + '.idn2.g'], P[d + '.idn2.b'], M=R, lda=2 * C,
                                      y=self.xn.view(-1)[C:], ldy=2 * C))
            o.append(self._lin(self.xn, d + '.self', self.sqvu, 2 * C, SQVU, relu=3, act_begin=D_ATT))
            o.append(ops.gated_attn(self.sqvu, self.sqvu, self.sqvu.view(-1)[D_ATT:], self.sqvu.view(-1)[D_ATT + E2:], self.g2, self._gp(),
                                    Lq=L, DV=E2, ldq=SQVU, ldk=SQVU, ldv=SQVU, ldua=SQVU, ldo=E2, nchunks=self_rows, frames=1, keys_per_frame=L,
                                    dw=P[d + '.self_dw.w'], H=self.H16, W=self.W16, nclips=B))
            o += self._tail(d + '.self', self.xc)
        for c in range(B):      # GroupNorm1D(512, 2) over fp32 rows, statistics per clip (760-808)
            o.append(ops.groupnorm(self._clip(self.xc, c, 2 * C), P['dec_gn.g'], P['dec_gn.b'], self._clip(self.dec_in, c, 2 * C),
                                   self.gn_ws, M=L, C=2 * C, groups=2))
        self._prog[key] = o
        return o

    def _write_memory(self, i: int) -> list:
        """Memory entry of the current frame of every clip for layer i into the window memory: K = curr_K, V = curr_V (pitched copies
        out of the fused GEMM's output), ID_V = SiLU(linear_ID_V([curr_ID_V | id_emb])) beside V (fuse_key_value_id, 1236-1242)."""
        R, C = self.B * self.L, D_MODEL
        k_dst, v_dst = self.short_K[i], self.short_V[i]
        o = [ops.copy2d_async(k_dst, D_ATT * 2, self.qvu[i], QVU * 2, D_ATT * 2, R),
             ops.copy2d_async(v_dst, E2 * 2, self.qvu[i].view(-1)[D_ATT:], QVU * 2, E1 * 2, R)]
        if i == 0:
            o.append(self._lin(self.id_emb, 'g0.idv', v_dst.view(-1)[E1:], C, E1, relu=3, ldo=E2))
        else:
            o.append(ops.copy2d_async(self.idcat[i].view(-1)[C:], 2 * C * 2, self.id_emb, C * 2, C * 2, R))
            o.append(self._lin(self.idcat[i], f'g{i}.idv', v_dst.view(-1)[E1:], 2 * C, E1, relu=3, ldo=E2))
        return o

    def _scatter_memory(self, i: int) -> list:
        """window memory of every clip -> the bank slot the device table names for it (negative = no append)."""
        L = self.L
        return [ops.scatter_blocks(self.short_K[i], self.bank_K[i], self.append_slots, nclips=self.B, block_bytes=L * D_ATT * 2,
                                   slot_bytes=L * D_ATT * 2),
                ops.scatter_blocks(self.short_V[i], self.bank_V[i], self.append_slots, nclips=self.B, block_bytes=L * E2 * 2,
                                   slot_bytes=L * E2 * 2)]

    def prog_id_emb(self, labels: torch.Tensor, hs: int, ws: int) -> list:
        """label maps -> one-hot -> identity bank conv -> LayerNorm (models/deaot.py:64-68), B clips at once."""
        key = f'id_{labels.data_ptr()}_{hs}_{ws}'
        if key in self._prog:
            return self._prog[key]
        P, B = self.P, self.B
        k, s, p = (17, 16, 8) if self.align else (16, 16, 0)
        o = [ops.label_to_onehot16(labels, self.onehot, Hs=hs, Ws=ws, Hd=self.H, Wd=self.W, ncls=self.nc, images=B),
             self._conv(self.onehot, P['idbank.w'], P['idbank.b'], self.id_raw, H=self.H, W=self.W, Cin=16, Cout=D_MODEL,
                        KH=k, KW=k, stride=s, pad=p),
             ops.layernorm256(self.id_raw, P['idnorm.g'], P['idnorm.b'], M=B * self.L, y=self.id_emb)]
        self._prog[key] = o
        return o

    def prog_update(self, append: bool) -> list:
        """update_short_memories / update_long_term_memory (transformer.py:825-872) for every clip: the window memory of every
        layer becomes the frame just propagated; with ``append`` the same entries join the banks."""
        key = f'update_{int(append)}'
        if key in self._prog:
            return self._prog[key]
        o = []
        for i in range(self.NL):
            o += self._write_memory(i)
            if append:
                o += self._scatter_memory(i)
        self._prog[key] = o
        return o
