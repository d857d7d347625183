"""ResNet-50 encoder over the frames of several clips at once.

Clips are independent but share the frozen encoder (encoders/resnet.py:10-196; models/aot.py:116-134), and one 481x849
frame gives GEMMs of only 1.7 k - 26 k rows.  ``BatchEncoder`` runs every encoder layer once for B frames (the conv
kernel's ``batch`` dimension: rows are [image][ho][wo]), writing the three stage outputs into [B, ...] buffers whose slice b
IS clip b's ``enc1 / enc2 / enc3`` (ClipRuntime.adopt_encoder_outputs), so each clip's own launch lists -- projector, LSTT,
decoder -- continue from there unchanged.  Results are bit-identical to the per-clip encoder: the same kernel computes
every output row from the same operands in the same order.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from . import ops
from .pack import R50_BLOCKS, R50_STRIDES
from .runtime import BF16, F32, _out


class BatchEncoder:
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
        self.img8 = e(B, H * W, 8)
        self.stem = e(B, self.H2 * self.W2, 64)
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
        o.append(ops.image_to_nhwc8(self.img_in, self.img8, H=self.H, W=self.W, images=B))
        o.append(self._conv(self.img8, P['stem.w'], P['stem.b'], self.stem, H=self.H, W=self.W, Cin=8, Cout=64, KH=7, KW=7, stride=2,
                            pad=3, relu=True))
        o.append(ops.maxpool3x3s2(self.stem, self.pool, H=self.H2, W=self.W2, C=64, images=B))
        x, (h, w), cin = self.pool, (self.H4, self.W4), 64
        outs = [self.x4, self.x8, self.x16]
        for li, (nblk, stride) in enumerate(zip(R50_BLOCKS, R50_STRIDES), start=1):
            planes = 64 * 2 ** (li - 1)
            for bi in range(nblk):
                p = f'encoder.layer{li}.{bi}'
                s = stride if bi == 0 else 1
                ho, wo = _out(h, 3, s, 1), _out(w, 3, s, 1)
                y = outs[li - 1][bi % 2]
                a = self.mid_a[: B * h * w * planes]
                bb = self.mid_b[: B * ho * wo * planes]
                o.append(self._conv(x, P[p + '.conv1.w'], P[p + '.conv1.b'], a, H=h, W=w, Cin=cin, Cout=planes, relu=True))
                o.append(self._conv(a, P[p + '.conv2.w'], P[p + '.conv2.b'], bb, H=h, W=w, Cin=planes, Cout=planes, KH=3, KW=3,
                                    stride=s, pad=1, relu=True))
                if (p + '.c3ds.w') in P:     # conv3 + strided 1x1 shortcut as one GEMM: the shortcut tensor never exists
                    o.append(ops.conv1x1_dual(bb, x, P[p + '.c3ds.w'], P[p + '.c3ds.b'], y, H=ho, W=wo, Cin=planes, Cout=planes * 4,
                                              H2=h, W2=w, Cin2=cin, stride2=s, relu=True, batch=B))
                else:
                    o.append(self._conv(bb, P[p + '.conv3.w'], P[p + '.conv3.b'], y, H=ho, W=wo, Cin=planes, Cout=planes * 4,
                                        residual=x, relu=True))
                x, (h, w), cin = y, (ho, wo), planes * 4
        self._prog = o
        return o
