"""The frozen image encoders over several frames at once (ResNet-50: BatchEncoder, Swin-B: SwinBatchEncoder).

Clips are independent but share the frozen encoder (encoders/resnet.py:10-196; models/aot.py:116-134), and one 481x849
frame gives GEMMs of only 1.7 k - 26 k rows.  ``BatchEncoder`` runs every encoder layer once for B frames (the conv
kernel's ``batch`` dimension: rows are [image][ho][wo]), writing the three stage outputs into [B, ...] buffers whose slice b
IS clip b's ``enc1 / enc2 / enc3`` (ClipRuntime.adopt_encoder_outputs), so each clip's own launch lists -- projector, LSTT,
decoder -- continue from there unchanged.  Results are bit-identical to the per-clip encoder: the same kernel computes
every output row from the same operands in the same order.
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import torch

from . import ops
from .pack import R50_BLOCKS, R50_STRIDES
from .runtime import BF16, F32, _out


class _PtrInput:
    """Input side of the batch encoders: the frames are named by a device table of pointers (one fp32 [3, H, W] frame each,
    wherever the caller keeps its clips) that set_frames() re-sends before a launch; img_in remains as a staging buffer for
    callers that copy their frames in (point_at_img_in: the default state)."""

    def _init_ptrs(self, B: int, device):
        self.img_ptrs = torch.tensor([self.img_in[i].data_ptr() for i in range(B)], dtype=torch.int64).to(device)
        self._ptr_ring = ops.PinnedRing(4, (B,), torch.int64, device)
        self._ptrs_set = False              # True: the table names caller-owned frames (set_frames), not img_in

    def _input_op(self):
        return ops.image_ptrs_to_nhwc8(self.img_ptrs, self.img8, H=self.H, W=self.W, images=self.B)

    def set_frames(self, frames, stream: int):
        """frames: B fp32 [3, H, W] contiguous device tensors (kept alive by the caller until the encoder launch has run)."""
        assert len(frames) == self.B
        host = self._ptr_ring.next()
        for i, f in enumerate(frames):
            assert f.dtype == F32 and f.is_contiguous() and f.numel() == 3 * self.H * self.W
            host[i] = f.data_ptr()
        self._ptr_ring.upload(self.img_ptrs, self.B * 8, stream)
        self._ptrs_set = True

    def point_at_img_in(self, stream: int):
        """the table names the rows of img_in again (callers that fill img_in themselves)"""
        if self._ptrs_set:
            self.set_frames([self.img_in[i] for i in range(self.B)], stream)
            self._ptrs_set = False


class BatchEncoder(_PtrInput):
    def __init__(self, P: Dict[str, torch.Tensor], in_hw: Tuple[int, int], batch: int, device):
        if 'pe.w' in P:
            raise ops.RmemError('BatchEncoder covers the ResNet-50 encoder')
        self.P, self.B, self.dev = P, batch, device
        H, W = in_hw
        self.H, self.W = H, W
        B = batch
        self.H2, self.W2 = _out(H, 7, 2, 3), _out(W, 7, 2, 3)
        self.H4, self.W4 = _out(self.H2, 3, 2, 1), _out(self.W2, 3, 2, 1)
        self.H8, self.W8 = _out(self.H4, 3, 2, 1), _out(self.W4, 3, 2, 1)
        self.H16, self.W16 = _out(self.H8, 3, 2, 1), _out(self.W8, 3, 2, 1)
        M4, M8, L = self.H4 * self.W4, self.H8 * self.W8, self.H16 * self.W16
        dt16 = P['stem.w'].dtype
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or dt16, device=device)  # noqa: E731
        self.img_in = e(B, 3, H, W, dt=F32)
        self._init_ptrs(B, device)
        # stem input: NHWC4 frames inside the zero border rmem_stem7x7s2 reads (zeroed once here, only the interior is ever written);
        # RMEM_STEM=rowrun keeps the 8-channel layout and the generic row-run GEMM form (experiments)
        self.stem4 = os.environ.get('RMEM_STEM', 'pool') != 'rowrun' and 'stem.w4' in P          # RMEM_STEM = pool (default) | direct | rowrun
        if self.stem4:
            hp, wp = ops.stem_padded_size(H, W)
            self.img4 = torch.zeros(B, hp, wp, 4, dtype=dt16, device=device)
        else:
            self.img8 = e(B, H * W, 8)
        self.stem_pool = self.stem4 and os.environ.get('RMEM_STEM', 'pool') == 'pool'
        self.stem = None if self.stem_pool else e(B, self.H2 * self.W2, 64)      # (stem + max-pool in one pass: the half-resolution map does not exist)
        self.pool = e(B, M4, 64)
        self.x4 = [e(B, M4, 256), e(B, M4, 256)]
        self.x8 = [e(B, M8, 512), e(B, M8, 512)]
        self.x16 = [e(B, L, 1024), e(B, L, 1024)]
        self.mid_a = e(B * M4 * 128)
        self.mid_b = e(B * M4 * 64)
        self.conv_ws = torch.empty(16 * B * L * 256, dtype=F32, device=device)
        self._prog = None
        # stage outputs = the last block's buffer of every layer (block count - 1) % 2
        self.enc_out = (self.x4[(R50_BLOCKS[0] - 1) % 2], self.x8[(R50_BLOCKS[1] - 1) % 2], self.x16[(R50_BLOCKS[2] - 1) % 2])

    def _conv(self, *a, **kw):
        return ops.conv2d(*a, ws=self.conv_ws, batch=self.B, **kw)

    def prog(self) -> list:
        if self._prog is not None:
            return self._prog
        P, B, o = self.P, self.B, []
        # RMEM_ENC_FRONT_SPLIT = n (experiment): stem and layer 1 run depth-first over n sub-batches of B / n frames (their maps are 211 MB
        # per 16 frames: a sub-batch's maps may stay in the Infinity Cache between producer and consumer), layers 2-3 over all B frames
        nsplit = int(os.environ.get('RMEM_ENC_FRONT_SPLIT', '1'))
        if nsplit < 1 or B % nsplit:
            nsplit = 1
        chain_on = os.environ.get('RMEM_NO_BNECK_CHAIN', '0') != '1'
        # layer 1 (64 channels); the 128-channel form (layer 2) is faster alone (57.3 -> 43.5 us) but takes a whole CU's LDS per workgroup
        # and measured neutral in the pipeline: opt-in (RMEM_DIRECT_CONV3_128=1)
        direct3 = () if os.environ.get('RMEM_NO_DIRECT_CONV3', '0') == '1' else ((64, 128) if os.environ.get('RMEM_DIRECT_CONV3_128') == '1' else (64,))
        blocks = [(li, bi) for li, nblk in enumerate(R50_BLOCKS, start=1) for bi in range(nblk)]
        outs = [self.x4, self.x8, self.x16]
        M4 = self.H4 * self.W4
        if nsplit > 1 and getattr(self, 'mid_a128', None) is None:
            self.mid_a128 = torch.empty(B * M4 * 128, dtype=self.pool.dtype, device=self.dev)     # layer 2's first conv1 output (see below)

        def sub(t, per_image, b0, nb):            # images b0 .. b0 + nb of a [B, ...] buffer stored image-major
            return t.reshape(-1)[b0 * per_image:(b0 + nb) * per_image]

        def stem(b0, nb):
            q = []
            conv = lambda *a_, **kw: ops.conv2d(*a_, ws=self.conv_ws, batch=nb, **kw)       # noqa: E731
            pool = sub(self.pool, M4 * 64, b0, nb)
            if self.stem4:
                hp, wp = ops.stem_padded_size(self.H, self.W)
                img4 = sub(self.img4, hp * wp * 4, b0, nb)
                q.append(ops.image_ptrs_to_nhwc4p(self.img_ptrs[b0:], img4, H=self.H, W=self.W, images=nb))
                if self.stem_pool:     # stem + max-pool in one pass: the half-resolution map is never written
                    q.append(ops.stem7x7s2_pool(img4, P['stem.w4'], P['stem.b'], pool, H=self.H, W=self.W, images=nb))
                else:
                    st = sub(self.stem, self.H2 * self.W2 * 64, b0, nb)
                    q.append(ops.stem7x7s2(img4, P['stem.w4'], P['stem.b'], st, H=self.H, W=self.W, images=nb))
                    q.append(ops.maxpool3x3s2(st, pool, H=self.H2, W=self.W2, C=64, images=nb))
            else:
                img8 = sub(self.img8, self.H * self.W * 8, b0, nb)
                st = sub(self.stem, self.H2 * self.W2 * 64, b0, nb)
                q.append(ops.image_ptrs_to_nhwc8(self.img_ptrs[b0:], img8, H=self.H, W=self.W, images=nb))
                q.append(conv(img8, P['stem.w'], P['stem.b'], st, H=self.H, W=self.W, Cin=8, Cout=64, KH=7, KW=7, stride=2, pad=3, relu=True))
                q.append(ops.maxpool3x3s2(st, pool, H=self.H2, W=self.W2, C=64, images=nb))
            return q

        def run_blocks(idxs, b0, nb, x, hw, cin, conv1_done):
            """the bottleneck blocks blocks[i], i in idxs, over images b0 .. b0 + nb; x: their input map (whole-batch buffer)"""
            q = []
            conv = lambda *a_, **kw: ops.conv2d(*a_, ws=self.conv_ws, batch=nb, **kw)       # noqa: E731
            h, w = hw
            for idx in idxs:
                li, bi = blocks[idx]
                stride = R50_STRIDES[li - 1]
                planes = 64 * 2 ** (li - 1)
                p = f'encoder.layer{li}.{bi}'
                s = stride if bi == 0 else 1
                ho, wo = _out(h, 3, s, 1), _out(w, 3, s, 1)
                xs = sub(x, h * w * cin, b0, nb)
                y = sub(outs[li - 1][bi % 2], ho * wo * planes * 4, b0, nb)
                # (layer 2's first conv1 output lives in its own buffer when the front is split: the 128-channel rows of sub-batch 0
                # would overlap the 64-channel rows of sub-batch 1 in mid_a)
                abuf = self.mid_a128 if (nsplit > 1 and li == 2 and bi == 0) else self.mid_a
                a = sub(abuf, h * w * planes, b0, nb)
                bb = sub(self.mid_b, ho * wo * planes, b0, nb)
                if not conv1_done:
                    q.append(conv(xs, P[p + '.conv1.w'], P[p + '.conv1.b'], a, H=h, W=w, Cin=cin, Cout=planes, relu=True))
                conv1_done = False
                if planes in direct3 and s == 1:     # read in place from rows kept in LDS, weights in registers (bit-identical)
                    q.append(ops.conv3x3_direct(a, P[p + '.conv2.w'], P[p + '.conv2.b'], bb, H=h, W=w, C=planes, images=nb, relu=True))
                else:
                    q.append(conv(a, P[p + '.conv2.w'], P[p + '.conv2.b'], bb, H=h, W=w, Cin=planes, Cout=planes, KH=3, KW=3, stride=s, pad=1,
                                  relu=True))
                dual = (p + '.c3ds.w') in P
                nxt = blocks[idx + 1] if idx + 1 < len(blocks) else None
                if chain_on and li == 1 and nxt is not None:
                    # layer 1 (256-channel maps at stride 4: HBM-bound at these batch sizes): a block's conv3 + shortcut is chained into the
                    # NEXT block's conv1 in one launch (rmem_bneck_chain): the 256-channel map is written once, not read back by that conv1
                    pn = f'encoder.layer{nxt[0]}.{nxt[1]}'
                    n2 = 64 * 2 ** (nxt[0] - 1)
                    nbuf = self.mid_a128 if (nsplit > 1 and nxt == (2, 0)) else self.mid_a
                    a_next = sub(nbuf, ho * wo * n2, b0, nb)
                    kw = dict(x2=xs, H2=h, W2=w, Cin2=cin, stride2=s) if dual else dict(residual=xs)
                    q.append(ops.bneck_chain(bb, P[p + ('.c3ds.w' if dual else '.conv3.w')], P[p + ('.c3ds.b' if dual else '.conv3.b')], y,
                                             P[pn + '.conv1.w'], P[pn + '.conv1.b'], a_next, H=ho, W=wo, K1=planes, N2=n2, batch=nb, **kw))
                    conv1_done = True
                elif dual:     # conv3 + strided 1x1 shortcut as one GEMM: the shortcut tensor never exists
                    q.append(ops.conv1x1_dual(bb, xs, P[p + '.c3ds.w'], P[p + '.c3ds.b'], y, H=ho, W=wo, Cin=planes, Cout=planes * 4,
                                              H2=h, W2=w, Cin2=cin, stride2=s, relu=True, batch=nb))
                else:
                    q.append(conv(bb, P[p + '.conv3.w'], P[p + '.conv3.b'], y, H=ho, W=wo, Cin=planes, Cout=planes * 4, residual=xs, relu=True))
                x, (h, w), cin = outs[li - 1][bi % 2], (ho, wo), planes * 4
            return q, x, (h, w), cin, conv1_done

        n1 = R50_BLOCKS[0]
        nb = B // nsplit
        for k in range(nsplit):
            o += stem(k * nb, nb)
            q, x, hw, cin, c1 = run_blocks(range(n1), k * nb, nb, self.pool, (self.H4, self.W4), 64, False)
            o += q
        q, x, hw, cin, c1 = run_blocks(range(n1, len(blocks)), 0, B, x, hw, cin, c1)
        o += q
        self._prog = o
        return o


class SwinBatchEncoder(_PtrInput):
    """Swin-B (cfg 5; encoders/swin/swin_transformer.py:500-716) over B frames: the look-ahead counterpart of
    ClipRuntime._prog_encode_swin.  One 720x1280 frame leaves only 3600 tokens for the 18 blocks of stage 3, so its linears are
    GEMMs of 3600 rows and its LayerNorms launches of 3.7 MB; with B frames stacked as rows [frame][token] every linear and
    LayerNorm is ONE launch over B times the rows (weights are shared, rows independent); window attention and patch merging
    depend on the image geometry and take the frame as a grid dimension.  Same kernels, same
    operands per row as the per-frame encoder; interface of BatchEncoder (img_in, prog(), enc_out)."""

    def __init__(self, P: Dict[str, torch.Tensor], in_hw: Tuple[int, int], batch: int, device):
        if 'pe.w' not in P:
            raise ops.RmemError('SwinBatchEncoder covers the Swin-B encoder')
        self.P, self.B, self.dev = P, batch, device
        H, W = in_hw
        if H % 4 or W % 4:
            raise ops.RmemError('Swin-B path: network size must be a multiple of 4')
        self.H, self.W = H, W
        B = batch
        self.H4, self.W4 = H // 4, W // 4
        self.H8, self.W8 = (self.H4 + 1) // 2, (self.W4 + 1) // 2
        self.H16, self.W16 = (self.H8 + 1) // 2, (self.W8 + 1) // 2
        M4, M8, L = self.H4 * self.W4, self.H8 * self.W8, self.H16 * self.W16
        dt16 = P['proj.w'].dtype
        e = lambda *shape, dt=None: torch.empty(*shape, dtype=dt or dt16, device=device)  # noqa: E731
        self.img_in = e(B, 3, H, W, dt=F32)
        self._init_ptrs(B, device)
        self.img8 = e(B, H * W, 8)
        self.sx = e(B * M4, 128, dt=F32)            # fp32 residual stream of the current stage
        self.sln = e(B * M4, 128)
        self.sqkv = e(B * M4, 384)
        self.satt = e(B * M4, 128)
        self.smlp = e(B * M4, 512)
        self.smerge = e(B * M8, 512)
        self.enc_out = (e(B, M4, 128), e(B, M8, 256), e(B, L, 512))
        self.conv_ws = torch.empty(16 * B * L * 256, dtype=F32, device=device)
        self._prog = None

    def _lin(self, x, name, y, M, K, N, **kw):
        return ops.linear(x, self.P[name + '.w'], self.P[name + '.b'], y, M=M, K=K, N=N, ws=self.conv_ws, **kw)

    def prog(self) -> list:
        if self._prog is not None:
            return self._prog
        from .pack import SWIN_DEPTHS, SWIN_HEADS
        P, B, o = self.P, self.B, []
        o.append(self._input_op())
        h, w, C = self.H4, self.W4, 128
        x = self.sx.view(-1)
        o.append(ops.conv2d(self.img8, P['pe.w'], P['pe.b'], x[: B * h * w * C], H=self.H, W=self.W, Cin=8, Cout=C, KH=4, KW=4, stride=4,
                            batch=B))
        o.append(ops.layernorm(x, P['pe.ln.g'], P['pe.ln.b'], M=B * h * w, C=C, yf=x))
        for li, (depth, heads) in enumerate(zip(SWIN_DEPTHS, SWIN_HEADS)):
            M = h * w
            ln, qkv, att, mlp = self.sln.view(-1), self.sqkv.view(-1), self.satt.view(-1), self.smlp.view(-1)
            for b in range(depth):
                d = f'sw{li}.{b}'
                o.append(ops.layernorm(x, P[d + '.norm1.g'], P[d + '.norm1.b'], M=B * M, C=C, y=ln))
                o.append(self._lin(ln, d + '.qkv', qkv, B * M, C, 3 * C))
                o.append(ops.window_attn(qkv, P[d + '.qkv.b'], P[d + '.table'], att, H=h, W=w, C=C, heads=heads,
                                         shift=0 if b % 2 == 0 else 3, images=B))      # windows are cut per image
                o.append(self._lin(att, d + '.proj', x, B * M, C, C, residual=x))
                o.append(ops.layernorm(x, P[d + '.norm2.g'], P[d + '.norm2.b'], M=B * M, C=C, y=ln))
                o.append(self._lin(ln, d + '.fc1', mlp, B * M, C, 4 * C, relu=2))
                o.append(self._lin(mlp, d + '.fc2', x, B * M, 4 * C, C, residual=x))
            o.append(ops.layernorm(x, P[f'sw.norm{li}.g'], P[f'sw.norm{li}.b'], M=B * M, C=C, y=self.enc_out[li]))
            if li < len(SWIN_DEPTHS) - 1:
                mg = self.smerge.view(-1)
                h2, w2 = (h + 1) // 2, (w + 1) // 2
                o.append(ops.patch_merge_ln(x, P[f'sw{li}.merge.g'], P[f'sw{li}.merge.b'], mg, H=h, W=w, C=C, images=B))
                h, w = h2, w2
                o.append(ops.linear(mg, P[f'sw{li}.merge.w'], None, x, M=B * h * w, K=4 * C, N=2 * C, ws=self.conv_ws))
                C *= 2
        self._prog = o
        return o
