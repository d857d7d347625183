"""Per-clip device state and the launch lists of one frame.

``ClipRuntime`` owns everything one AOTEngine needs on the GPU for one network size:
activation buffers (NHWC bf16; the LSTT residual stream is fp32), the long-term memory
bank as a slot ring per layer (bf16 [slots, HW, 256] for K and V, an index table instead
of the reference's torch.cat / slice, layers/transformer.py:319, 432-433), the short-term
memory, and the prepared launch lists (``ops.Op``) for

    encode      image -> ResNet-50 -> 1x1 projector            (models/aot.py:116-134)
    lstt_ref    3 LSTT layers, reference-frame mode            (layers/transformer.py:582-588)
    lstt_prop   3 LSTT layers, propagate mode, bank size T     (layers/transformer.py:589-692)
    decode      FPN head -> logits at 1/4 resolution           (decoders/fpn.py:36-68)
    id_emb      label map -> one-hot -> identity bank conv     (engines/aot_engine.py:208-232)
    update      short/long-term memory update                  (layers/transformer.py:269-322)

The lists are pure functions of the buffers' addresses, so each one is built once and
replayed (directly or as a captured hipGraph).  Everything that changes from frame to
frame (slot table, temporal-PE slots) lives in device memory.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from .pack import R50_BLOCKS, R50_STRIDES

BF16, F32 = torch.bfloat16, torch.float32
D_MODEL, HEADS, FFN = 256, 8, 1024
MAX_CHUNKS = 32
# key ranges of the one-frame (self / short-term) attention launches: 1 = every workgroup walks all keys and writes the
# normalised output itself (no partials, no combine launch)
PLAIN_CHUNKS = int(__import__('os').environ.get('RMEM_PLAIN_CHUNKS', 1))


def temporal_slots(T: int, n_slots: int = 4) -> List[int]:
    """Temporal-PE slot of each bank entry (layers/transformer.py:598-621).

    T <= 4: entry t uses slot t.  T > 4: the slots are flipped, nearest-resized to T
    (src = floor(dst * float32(4 / T))) and flipped back.
    """
    if T <= n_slots:
        return list(range(T))
    scale = torch.tensor(float(n_slots), dtype=F32) / torch.tensor(float(T), dtype=F32)
    out = []
    for t in range(T):
        src = int(torch.floor(torch.tensor(float(T - 1 - t), dtype=F32) * scale).item())
        out.append(n_slots - 1 - min(src, n_slots - 1))
    return out


def sine_pos_emb(h: int, w: int, c: int = D_MODEL) -> torch.Tensor:
    """2-D sine positional embedding [h*w, c] (layers/position.py:50-77: normalize=True,
    scale 2*pi, temperature 1e4, y half then x half); computed once per clip on the host."""
    nf = c // 2
    ys = torch.arange(h, dtype=F32)[:, None].expand(h, w)
    xs = torch.arange(w, dtype=F32)[None, :].expand(h, w)
    ys = ys / (ys[-1:, :] + 1e-6) * (2 * torch.pi)
    xs = xs / (xs[:, -1:] + 1e-6) * (2 * torch.pi)
    dim_t = 10000 ** (2 * torch.div(torch.arange(nf, dtype=F32), 2, rounding_mode='trunc') / nf)
    px, py = xs[:, :, None] / dim_t, ys[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    return torch.cat((py, px), dim=2).reshape(h * w, c).contiguous()


def _out(n, k, s, p):
    return (n + 2 * p - k) // s + 1


class ClipRuntime:
    max_chunks = MAX_CHUNKS
    bank_kw, bank_vw = D_MODEL, D_MODEL      # row widths of the bank's K and V entries

    def __init__(self, P: Dict[str, torch.Tensor], in_hw: Tuple[int, int], bank_slots: int, device,
                 num_lstt: int = 3, align_corners: bool = True, num_classes: int = 11):
        self.P, self.dev, self.NL = P, device, num_lstt
        self.dt = P['proj.w'].dtype              # 16-bit element type of activations and bank = that of the packed weights
        self.align = align_corners
        self.nc = num_classes
        H, W = in_hw
        self.H, self.W = H, W
        self.swin = 'pe.w' in P
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or self.dt, device=device)  # noqa: E731
        self.img8 = e(H * W, 8)
        if self.swin:
            # Swin-B (cfg 5): patch 4, then two patch mergings (encoders/swin/swin_transformer.py:500-545, 684-716)
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
        L = self.L
        M4, M8 = self.H4 * self.W4, self.H8 * self.W8

        # ---- encoder buffers ----
        if self.swin:
            self.sx = e(M4, 128, dt=F32)            # fp32 residual stream of the current stage (M4*128 >= M8*256 >= L*512)
            self.sln = e(M4, 128)                   # LayerNorm output (bf16)
            self.sqkv = e(M4, 384)
            self.satt = e(M4, 128)
            self.smlp = e(M4, 512)
            self.smerge = e(M8, 512)                # patch-merge LN output [tokens/4, 4C]
            self.enc1, self.enc2, self.enc3 = e(M4, 128), e(M8, 256), e(L, 512)
        else:
            self.stem = e(self.H2 * self.W2, 64)
            self.pool = e(M4, 64)
            self.x4 = [e(M4, 256), e(M4, 256)]          # layer1 ping-pong
            self.x8 = [e(M8, 512), e(M8, 512)]
            self.x16 = [e(L, 1024), e(L, 1024)]
            self.mid_a = e(M4, 128)                     # bottleneck conv1 out (<= M4*64, M4*128 for layer2.0, ...)
            self.mid_b = e(M4, 64)                      # bottleneck conv2 out
        self._alloc_lstt(L, num_lstt)
        self.onehot = e(H * W, 16)
        self.gn_ws = ops.groupnorm_workspace(32, device)
        self.conv_ws = torch.empty(16 * L * D_MODEL, dtype=F32, device=device)      # split-K slabs (<= 16 slices of [HW, 256])
        self.mass = torch.zeros(L, self.max_chunks, dtype=F32, device=device)
        self.scores = torch.zeros(32 + 64 * 32, dtype=F32, device=device)     # T scores + reduction scratch
        # ---- decoder buffers ----
        self.d16a, self.d16b = e(L, 256), e(L, 256)
        self.d8a, self.d8b = e(M8, 256), e(M8, 256)
        self.d4a, self.d4b = e(M4, 128), e(M4, 128)
        self.logits = torch.zeros(M4, 16, dtype=F32, device=device)
        # ---- memory bank ----
        self.chunks = torch.zeros(self.max_chunks, 8, dtype=torch.int32, device=device)
        self.chunks_ring = ops.PinnedRing(4, (self.max_chunks, 8), torch.int32, device)
        self.scores_host = torch.zeros(MAX_CHUNKS, dtype=F32).pin_memory()
        self.bank_generation = 0             # bumped when the bank is re-allocated: launch lists / graphs built on it are stale
        self._alloc_bank(bank_slots)
        self._prog: Dict[str, list] = {}

    def _alloc_lstt(self, L: int, num_lstt: int):
        """Activation buffers of the propagation stack (AOT: 3 LSTT blocks)."""
        device = self.dev
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or self.dt, device=device)  # noqa: E731
        self.x = e(L, D_MODEL, dt=F32)              # residual stream
        self.dec_in = e(L, 4 * D_MODEL)             # cat(enc256, 3 x normed LSTT out), decoders/fpn.py:38-39
        self.t1b, self.t1p = e(L, D_MODEL), e(L, D_MODEL)
        self.qkv = e(L, 3 * D_MODEL)
        self.att = e(L, D_MODEL)
        self.t3 = e(L, D_MODEL)
        self.k4, self.v4 = e(L, D_MODEL), e(L, D_MODEL)
        self.h1, self.h2, self.h3 = e(L, FFN), e(L, FFN), e(L, FFN)
        self.tmp = e(L, D_MODEL)
        self.curr_Q = [e(L, D_MODEL) for _ in range(num_lstt)]   # = curr_K
        self.curr_V = [e(L, D_MODEL) for _ in range(num_lstt)]   # LN2 output
        self.tgt3 = [e(L, D_MODEL) for _ in range(num_lstt)]
        self.short_K = [e(L, D_MODEL) for _ in range(num_lstt)]
        self.short_V = [e(L, D_MODEL) for _ in range(num_lstt)]
        self.id_emb = e(L, D_MODEL)
        self.pos = sine_pos_emb(self.H16, self.W16).to(device)
        # pos @ [Wq; Wk]^T per layer (fp32, V columns zero): the residual operand of the fused self-attention QKV GEMM
        self.pos_qk = [torch.zeros(L, 3 * D_MODEL, dtype=F32, device=device) for _ in range(num_lstt)]
        self._pos_ready = False
        self.attn_ws = ops.attn_workspace(L, HEADS, MAX_CHUNKS, device)
        self.dec_cin = 4 * D_MODEL

    # ------------------------------------------------------------------ bank
    def _alloc_bank(self, slots: int):
        self.S = slots
        self.bank_K = [torch.empty(slots, self.L, self.bank_kw, dtype=self.dt, device=self.dev) for _ in range(self.NL)]
        self.bank_V = [torch.empty(slots, self.L, self.bank_vw, dtype=self.dt, device=self.dev) for _ in range(self.NL)]
        self.slots: List[int] = []           # logical order t -> physical slot
        self.free: List[int] = list(range(slots))
        self._on_bank_resized()

    def _on_bank_resized(self):
        pass

    def reset_bank(self):
        self.slots = []
        self.free = list(range(self.S))

    def grow_bank(self):
        """Unbounded-memory mode (latter_mem_len = 9999, tools/eval.py:92): double the ring."""
        old_K, old_V, old_S = self.bank_K, self.bank_V, self.S
        new_S = old_S * 2
        self.bank_K = [torch.empty(new_S, self.L, self.bank_kw, dtype=self.dt, device=self.dev) for _ in range(self.NL)]
        self.bank_V = [torch.empty(new_S, self.L, self.bank_vw, dtype=self.dt, device=self.dev) for _ in range(self.NL)]
        for i in range(self.NL):
            self.bank_K[i][:old_S].copy_(old_K[i])
            self.bank_V[i][:old_S].copy_(old_V[i])
        self.free += list(range(old_S, new_S))
        self.S = new_S
        self._on_bank_resized()
        self.bank_generation += 1
        self._prog = {k: v for k, v in self._prog.items() if not k.startswith(('lstt', 'update'))}

    def take_slot(self) -> int:
        if not self.free:
            self.grow_bank()
        return self.free.pop(0)

    def chunk_plan(self, T: int) -> Tuple[int, int]:
        """(splits per memory frame, chunk count): keep >= ~8 key chunks in flight for small T."""
        if T > MAX_CHUNKS:
            raise ops.RmemError(f'memory bank of {T} frames exceeds the {MAX_CHUNKS}-chunk table')
        splits = max(1, min(8 // T, MAX_CHUNKS // T))
        return splits, T * splits

    def _keys_per_chunk(self, splits: int) -> int:
        return (self.L + splits - 1) // splits

    def _chunk_rows(self, slots: List[int]):
        T = len(slots)
        splits, n = self.chunk_plan(T)
        pes = temporal_slots(T)
        per = self._keys_per_chunk(splits)
        rows = []
        for t, s in enumerate(slots):
            for kb in range(0, self.L, per):
                rows.append((s, kb, min(per, self.L - kb), pes[t], t))
        assert len(rows) == n
        return rows, n

    def mem_read_probe(self, T: int, layer: int = 0):
        """Stand-alone Op of the long-term memory read at bank size T (bench.py's roofline leg; see GroupRuntime.mem_read_probe)."""
        if not 1 <= T <= self.S:
            raise ops.RmemError(f'mem_read_probe: T = {T} outside 1..{self.S}')
        rows, n = self._chunk_rows(list(range(T)))
        table = torch.zeros(n, 8, dtype=torch.int32)
        table[:, :5] = torch.tensor(rows, dtype=torch.int32)
        self._probe_chunks = table.to(self.dev)
        L, C, i = self.L, D_MODEL, layer
        op = self._attn(self.curr_Q[i], C, self.bank_K[i], self.bank_V[i], C, self.att, slot_stride=L * C, chunks=self._probe_chunks,
                        nchunks=n, lk_single=T * L, pe_cur=self.P['pe_cur'], pe_mem=self.P['pe_mem'], mass=None, T=T)
        return op, 4.0 * L * (T * L) * C

    def upload_chunks(self, stream: int):
        """Write the chunk table for the current slot order (call after every bank change)."""
        rows, n = self._chunk_rows(self.slots)
        # a pinned staging row whose previous upload has executed (ops.PinnedRing waits for it if it has not)
        host = self.chunks_ring.next()
        host.zero_()
        host[:n, :5] = torch.tensor(rows, dtype=torch.int32)
        self.chunks_ring.upload(self.chunks, self.max_chunks * 8 * 4, stream)

    # ------------------------------------------------------------------ programs
    def _conv(self, *a, **kw):
        return ops.conv2d(*a, ws=self._ws(), **kw)

    def _ws(self):
        import os
        return None if os.environ.get('RMEM_NO_SPLITK') else self.conv_ws

    def _lin(self, x, name, y, M, K, N, **kw):
        return ops.linear(x, self.P[name + '.w'], self.P[name + '.b'], y, M=M, K=K, N=N, ws=self._ws(), **kw)

    def _prog_encode_swin(self, img: torch.Tensor) -> list:
        """Swin-B: patch embed + LN, 3 stages of (shifted-)window blocks, patch merging, per-stage output norms."""
        from .pack import SWIN_DEPTHS, SWIN_HEADS
        P, o = self.P, []
        o.append(ops.image_to_nhwc8(img, self.img8, H=self.H, W=self.W))
        h, w, C = self.H4, self.W4, 128
        x = self.sx.view(-1)
        o.append(ops.conv2d(self.img8, P['pe.w'], P['pe.b'], x[: h * w * C], H=self.H, W=self.W, Cin=8, Cout=C, KH=4, KW=4, stride=4))
        o.append(ops.layernorm(x, P['pe.ln.g'], P['pe.ln.b'], M=h * w, C=C, yf=x))
        outs = (self.enc1, self.enc2, self.enc3)
        for li, (depth, heads) in enumerate(zip(SWIN_DEPTHS, SWIN_HEADS)):
            M = h * w
            ln, qkv, att, mlp = self.sln.view(-1), self.sqkv.view(-1), self.satt.view(-1), self.smlp.view(-1)
            for b in range(depth):
                d = f'sw{li}.{b}'
                o.append(ops.layernorm(x, P[d + '.norm1.g'], P[d + '.norm1.b'], M=M, C=C, y=ln))
                o.append(self._lin(ln, d + '.qkv', qkv, M, C, 3 * C))
                o.append(ops.window_attn(qkv, P[d + '.qkv.b'], P[d + '.table'], att, H=h, W=w, C=C, heads=heads, shift=0 if b % 2 == 0 else 3))
                o.append(self._lin(att, d + '.proj', x, M, C, C, residual=x))
                o.append(ops.layernorm(x, P[d + '.norm2.g'], P[d + '.norm2.b'], M=M, C=C, y=ln))
                o.append(self._lin(ln, d + '.fc1', mlp, M, C, 4 * C, relu=2))
                o.append(self._lin(mlp, d + '.fc2', x, M, 4 * C, C, residual=x))
            o.append(ops.layernorm(x, P[f'sw.norm{li}.g'], P[f'sw.norm{li}.b'], M=M, C=C, y=outs[li]))
            if li < len(SWIN_DEPTHS) - 1:
                mg = self.smerge.view(-1)
                o.append(ops.patch_merge_ln(x, P[f'sw{li}.merge.g'], P[f'sw{li}.merge.b'], mg, H=h, W=w, C=C))
                h, w = (h + 1) // 2, (w + 1) // 2
                o.append(ops.linear(mg, P[f'sw{li}.merge.w'], None, x, M=h * w, K=4 * C, N=2 * C, ws=self.conv_ws))
                C *= 2
        o.append(self._proj_op(512))
        return o

    def prog_encode(self, img: torch.Tensor) -> list:
        """img: fp32 [3, H, W] device tensor at a FIXED address (the caller copies frames into it)."""
        key = 'encode'
        if key in self._prog:
            return self._prog[key]
        if self.swin:
            self._prog[key] = self._prog_encode_swin(img)
            return self._prog[key]
        P, o = self.P, []
        o.append(ops.image_to_nhwc8(img, self.img8, H=self.H, W=self.W))
        o.append(self._conv(self.img8, P['stem.w'], P['stem.b'], self.stem, H=self.H, W=self.W, Cin=8, Cout=64, KH=7, KW=7,
                            stride=2, pad=3, relu=True))
        o.append(ops.maxpool3x3s2(self.stem, self.pool, H=self.H2, W=self.W2, C=64))
        x, (h, w), cin = self.pool, (self.H4, self.W4), 64
        outs = [self.x4, self.x8, self.x16]
        for li, (nblk, stride) in enumerate(zip(R50_BLOCKS, R50_STRIDES), start=1):
            planes = 64 * 2 ** (li - 1)
            for bi in range(nblk):
                p = f'encoder.layer{li}.{bi}'
                s = stride if bi == 0 else 1
                ho, wo = _out(h, 3, s, 1), _out(w, 3, s, 1)
                y = outs[li - 1][bi % 2]
                a = self.mid_a.view(-1)[: h * w * planes]
                b = self.mid_b.view(-1)[: ho * wo * planes]
                o.append(self._conv(x, P[p + '.conv1.w'], P[p + '.conv1.b'], a, H=h, W=w, Cin=cin, Cout=planes, relu=True))
                o.append(self._conv(a, P[p + '.conv2.w'], P[p + '.conv2.b'], b, H=h, W=w, Cin=planes, Cout=planes, KH=3, KW=3,
                                    stride=s, pad=1, relu=True))
                if (p + '.c3ds.w') in P:     # conv3 + strided 1x1 shortcut as one GEMM: the shortcut tensor never exists
                    o.append(ops.conv1x1_dual(b, x, P[p + '.c3ds.w'], P[p + '.c3ds.b'], y, H=ho, W=wo, Cin=planes, Cout=planes * 4,
                                              H2=h, W2=w, Cin2=cin, stride2=s, relu=True))
                else:
                    o.append(self._conv(b, P[p + '.conv3.w'], P[p + '.conv3.b'], y, H=ho, W=wo, Cin=planes, Cout=planes * 4,
                                        residual=x, relu=True))
                x, (h, w), cin = y, (ho, wo), planes * 4
            setattr(self, f'enc{li}', x)
        # encoder_projector: fp32 residual stream + bf16 copy into the decoder's concat buffer
        o.append(self._proj_op(1024))
        self._prog[key] = o
        return o

    def _proj_op(self, cin: int, enc3=None):
        """encoder_projector (models/aot.py:25-29): fp32 residual stream + bf16 copy into the decoder's concat buffer."""
        return self._conv(self.enc3 if enc3 is None else enc3, self.P['proj.w'], self.P['proj.b'], self.x, H=self.L, W=1, Cin=cin,
                          Cout=D_MODEL, y2=self.dec_in, ld2=4 * D_MODEL)

    # ------------------------------------------------------------------ encoder look-ahead
    def batch_encoder(self, frames: int):
        """The frames of a clip do not depend on each other before the LSTT, so the encoder (ResNet-50 or Swin-B) may run
        ``frames`` frames ahead as ONE launch per layer (rmem_ocu_amd.encoder_batch); slot e of its outputs then feeds
        prog_project(e) / prog_decode(e) of the frame that is propagated."""
        if getattr(self, '_benc', None) is None or self._benc.B != frames:
            from .encoder_batch import BatchEncoder, SwinBatchEncoder
            self._benc = (SwinBatchEncoder if self.swin else BatchEncoder)(self.P, (self.H, self.W), frames, self.dev)
            self._prog = {k: v for k, v in self._prog.items() if not k.startswith(('project_', 'decode_'))}
        return self._benc

    def _enc(self, e):
        """(enc1, enc2, enc3) of the frame in flight: the runtime's own encoder buffers, or slot e of the look-ahead batch."""
        if e is None:
            return self.enc1, self.enc2, self.enc3
        o = self._benc.enc_out
        return o[0][e], o[1][e], o[2][e]

    def prog_project(self, e: int) -> list:
        key = f'project_{e}'
        if key not in self._prog:
            self._prog[key] = [self._proj_op(self.enc_ch[2], enc3=self._enc(e)[2])]
        return self._prog[key]

    def _attn(self, q, ldq, k, v, ldkv, out, **kw):
        return ops.mem_read_attn(q, k, v, out, self.attn_ws, Lq=self.L, heads=HEADS, ldq=ldq, ldkv=ldkv, ldo=D_MODEL, **kw)

    def prepare_pos(self, stream: int):
        """One-off per runtime: pos_qk[i][:, :512] = bf16(pos) @ [Wq; Wk]^T (no bias; the QKV GEMM adds it)."""
        if self._pos_ready:
            return
        posb = self.pos.to(self.dt)
        for i in range(self.NL):
            ops.run(ops.linear(posb, self.P[f'l{i}.self_qk.w'], None, self.pos_qk[i], M=self.L, K=D_MODEL, N=2 * D_MODEL,
                               ldo=3 * D_MODEL), stream)
        self._keep_posb = posb
        self._pos_ready = True

    def prog_lstt(self, ref_mode: bool, T: int, ref_slot: int = 0, want_mass: bool = True) -> list:
        """The 3-layer LSTT on self.x.  ref_mode: reference frame (id_emb already in self.id_emb,
        K/V go straight into bank slot ``ref_slot``); else propagate against a bank of T frames."""
        key = f'lstt_ref{ref_slot}' if ref_mode else f'lstt_prop{T}{"m" if want_mass else ""}'
        if key in self._prog:
            return self._prog[key]
        P, L, o = self.P, self.L, []
        C = D_MODEL
        _, nchunks = self.chunk_plan(1 if ref_mode else T)
        for i in range(self.NL):
            d = f'l{i}'
            # --- self attention (transformer.py:565-571)
            o.append(ops.layernorm256(self.x, P[d + '.ln1.g'], P[d + '.ln1.b'], M=L, y=self.t1b))
            o.append(self._lin(self.t1b, d + '.self_qkv', self.qkv, L, C, 3 * C, residual=self.pos_qk[i]))
            o.append(self._attn(self.qkv, 3 * C, self.qkv.view(-1)[C:], self.qkv.view(-1)[2 * C:], 3 * C, self.att,
                                nchunks=PLAIN_CHUNKS, lk_single=L))
            o.append(self._lin(self.att, d + '.self_proj', self.x, L, C, C, residual=self.x))
            # --- long/short-term attention (573-680)
            o.append(ops.layernorm256(self.x, P[d + '.ln2.g'], P[d + '.ln2.b'], M=L, y=self.curr_V[i]))
            if ref_mode:
                cq = self.bank_K[i][ref_slot]                       # curr_K is the bank's first entry
                gv = self.bank_V[i][ref_slot]
                o.append(self._lin(self.curr_V[i], d + '.linear_Q', cq, L, C, C))
                o.append(ops.add16(self.curr_V[i], self.id_emb, self.tmp, L * C))
                o.append(self._lin(self.tmp, d + '.linear_V', gv, L, C, C))
                sk, sv = cq, gv                                     # local_K/V = global_K/V (585-586)
            else:
                cq = self.curr_Q[i]
                o.append(self._lin(self.curr_V[i], d + '.linear_Q', cq, L, C, C))
                sk, sv = self.short_K[i], self.short_V[i]
            o.append(self._attn(cq, C, self.bank_K[i], self.bank_V[i], C, self.att, slot_stride=L * C, chunks=self.chunks,
                                nchunks=nchunks, lk_single=(1 if ref_mode else T) * L, pe_cur=P['pe_cur'], pe_mem=P['pe_mem'],
                                mass=self.mass if (i == 0 and not ref_mode and want_mass) else None, T=T))
            o.append(self._lin(self.att, d + '.long_proj', self.x, L, C, C, residual=self.x))
            o.append(ops.layernorm256_pair(sk, cq, self.k4, sv, self.curr_V[i], self.v4, P[d + '.ln4.g'], P[d + '.ln4.b'], M=L))
            o.append(self._attn(cq, C, self.k4, self.v4, C, self.att, nchunks=PLAIN_CHUNKS, lk_single=L))
            o.append(self._lin(self.att, d + '.short_proj', self.x, L, C, C, residual=self.x, y2=self.tgt3[i]))
            if ref_mode:   # short-term memory of the reference frame (675-678)
                o.append(self._lin(self.tgt3[i], d + '.linear_QMem', self.short_K[i], L, C, C))
                o.append(ops.add16(self.tgt3[i], self.id_emb, self.tmp, L * C))
                o.append(self._lin(self.tmp, d + '.linear_VMem', self.short_V[i], L, C, C))
            # --- feed-forward (683-687)
            o.append(ops.layernorm256(self.x, P[d + '.ln3.g'], P[d + '.ln3.b'], M=L, y=self.t3))
            o.append(self._lin(self.t3, d + '.linear1', self.h1, L, C, FFN))
            o.append(ops.gn_act_dwconv5x5(self.h1, P[d + '.gn.g'], P[d + '.gn.b'], P[d + '.dw.w'], self.h3, self.gn_ws, H=self.H16,
                                          W=self.W16, C=FFN, groups=32, act=2))
            o.append(self._lin(self.h3, d + '.linear2', self.x, L, FFN, C, residual=self.x))
            # --- decoder norm of this layer's output into the concat buffer (248-259)
            o.append(ops.layernorm256(self.x, P[f'dec_norm{i}.g'], P[f'dec_norm{i}.b'], M=L,
                                      y=self.dec_in.view(-1)[(i + 1) * C:], ldy=4 * C))
        self._prog[key] = o
        return o

    def prog_decode(self, e=None) -> list:
        key = 'decode' if e is None else f'decode_{e}'
        if key in self._prog:
            return self._prog[key]
        P, o, L = self.P, [], self.L
        enc1, enc2, enc3 = self._enc(e)
        M8, M4 = self.H8 * self.W8, self.H4 * self.W4
        gn = lambda x, name, y, M, C: ops.groupnorm(x, P[name + '.gn.g'], P[name + '.gn.b'], y, self.gn_ws, M=M, C=C, groups=8, act=1)  # noqa: E731
        o.append(self._conv(self.dec_in, P['dec.conv_in.w'], P['dec.conv_in.b'], self.d16a, H=L, W=1, Cin=self.dec_cin, Cout=256))
        o.append(gn(self.d16a, 'dec.conv_in', self.d16b, L, 256))
        c4, c8, c16 = self.enc_ch
        o.append(self._conv(enc3, P['dec.adapter_16x.w'], P['dec.adapter_16x.b'], self.d16a, H=L, W=1, Cin=c16, Cout=256,
                            residual=self.d16b))
        o.append(self._conv(self.d16a, P['dec.conv_16x.w'], P['dec.conv_16x.b'], self.d16b, H=self.H16, W=self.W16, Cin=256, Cout=256,
                            KH=3, KW=3, pad=1))
        o.append(gn(self.d16b, 'dec.conv_16x', self.d16a, L, 256))
        import os
        fuse_up = not os.environ.get('RMEM_NO_UPFUSE')      # timing experiments only
        # F.interpolate(x, size) + adapter(shortcut) (decoders/fpn.py:49-52): the resize happens in the GEMM's residual read
        if fuse_up:
            o.append(self._conv(enc2, P['dec.adapter_8x.w'], P['dec.adapter_8x.b'], self.d8b, H=self.H8, W=self.W8, Cin=c8, Cout=256,
                                residual=self.d16a, res_up=(self.H16, self.W16, self.align)))
        else:
            o.append(ops.bilinear(self.d16a, self.d8a, Hi=self.H16, Wi=self.W16, Ho=self.H8, Wo=self.W8, C=256, align_corners=self.align))
            o.append(self._conv(enc2, P['dec.adapter_8x.w'], P['dec.adapter_8x.b'], self.d8b, H=M8, W=1, Cin=c8, Cout=256,
                                residual=self.d8a))
        d8c = self.d8a.view(-1)[: M8 * 128]
        o.append(self._conv(self.d8b, P['dec.conv_8x.w'], P['dec.conv_8x.b'], d8c, H=self.H8, W=self.W8, Cin=256, Cout=128,
                            KH=3, KW=3, pad=1))
        d8d = self.d8b.view(-1)[: M8 * 128]
        o.append(gn(d8c, 'dec.conv_8x', d8d, M8, 128))
        if fuse_up:
            o.append(self._conv(enc1, P['dec.adapter_4x.w'], P['dec.adapter_4x.b'], self.d4b, H=self.H4, W=self.W4, Cin=c4, Cout=128,
                                residual=d8d, res_up=(self.H8, self.W8, self.align)))
        else:
            o.append(ops.bilinear(d8d, self.d4a, Hi=self.H8, Wi=self.W8, Ho=self.H4, Wo=self.W4, C=128, align_corners=self.align))
            o.append(self._conv(enc1, P['dec.adapter_4x.w'], P['dec.adapter_4x.b'], self.d4b, H=M4, W=1, Cin=c4, Cout=128,
                                residual=self.d4a))
        o.append(self._conv(self.d4b, P['dec.conv_4x.w'], P['dec.conv_4x.b'], self.d4a, H=self.H4, W=self.W4, Cin=128, Cout=128,
                            KH=3, KW=3, pad=1))
        if os.environ.get('RMEM_NO_HEADFUSE'):               # timing experiments only
            o.append(gn(self.d4a, 'dec.conv_4x', self.d4b, M4, 128))
            o.append(self._conv(self.d4b, P['dec.conv_out.w'], P['dec.conv_out.b'], self.logits, H=M4, W=1, Cin=128, Cout=self.nc, ldo=16))
        else:                                                # conv_out(relu(gn(x))) in one pass over x (decoders/fpn.py:62-66)
            o.append(ops.groupnorm_head(self.d4a, P['dec.conv_4x.gn.g'], P['dec.conv_4x.gn.b'], P['dec.conv_out.w'], P['dec.conv_out.b'],
                                        self.logits, self.gn_ws, M=M4, C=128, groups=8, N=self.nc, ldy=16, act=1))
        self._prog[key] = o
        return o

    def prog_id_emb(self, label: torch.Tensor, hs: int, ws: int) -> list:
        """label: uint8 or fp32 [hs, ws] device tensor at a fixed address -> self.id_emb."""
        key = f'id_{label.data_ptr()}_{hs}_{ws}'
        if key in self._prog:
            return self._prog[key]
        P = self.P
        k, s, p = (17, 16, 8) if self.align else (16, 16, 0)
        o = [ops.label_to_onehot16(label, self.onehot, Hs=hs, Ws=ws, Hd=self.H, Wd=self.W, ncls=self.nc),
             self._conv(self.onehot, P['idbank.w'], P['idbank.b'], self.id_emb, H=self.H, W=self.W, Cin=16, Cout=D_MODEL,
                        KH=k, KW=k, stride=s, pad=p)]
        self._prog[key] = o
        return o

    def prog_update(self, append_slot: Optional[int]) -> list:
        """Memory update after a propagated frame (layers/transformer.py:269-322); self.id_emb holds the
        identity embedding of the predicted mask.  append_slot: bank slot receiving (curr_K, linear_V(curr_V + id))."""
        key = f'update_{append_slot}'
        if key in self._prog:
            return self._prog[key]
        L, C, o = self.L, D_MODEL, []
        P, NL = self.P, self.NL
        if NL > 4:
            raise ops.RmemError('memory update: grouped launches cover up to 4 LSTT layers')
        if not hasattr(self, 'tmpA'):
            self.tmpA = [torch.empty(L, C, dtype=self.dt, device=self.dev) for _ in range(NL)]
            self.tmpB = [torch.empty(L, C, dtype=self.dt, device=self.dev) for _ in range(NL)]
        app = append_slot is not None
        # the three layers' updates are independent: one launch per kind of op for all layers
        o.append(ops.add16_grouped(self.tgt3 + (self.curr_V if app else []), [self.id_emb] * (NL * (2 if app else 1)),
                                      self.tmpA + (self.tmpB if app else []), L * C))
        w = lambda nm: [P[f'l{i}.{nm}.w'] for i in range(NL)]   # noqa: E731
        b = lambda nm: [P[f'l{i}.{nm}.b'] for i in range(NL)]   # noqa: E731
        o.append(ops.linear_grouped(self.tgt3, w('linear_QMem'), b('linear_QMem'), self.short_K, M=L, K=C, N=C))
        o.append(ops.linear_grouped(self.tmpA, w('linear_VMem'), b('linear_VMem'), self.short_V, M=L, K=C, N=C))
        if app:
            o.append(ops.linear_grouped(self.tmpB, w('linear_V'), b('linear_V'), [self.bank_V[i][append_slot] for i in range(NL)],
                                        M=L, K=C, N=C))
            for i in range(NL):
                o.append(ops.copy_async(self.bank_K[i][append_slot], self.curr_Q[i], L * C * 2))
        self._prog[key] = o
        return o
