"""CPU oracle: fp32 restatement of the RMem / AOT-L per-frame inference path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``rmem_ocu_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the timed baseline.

What it restates (all citations relative to /root/reference/aot_plus/):
  * networks/encoders/resnet.py:10-68, 178-195   ResNet-50 stem + layer1..3
  * networks/layers/normalization.py:35-43       FrozenBatchNorm2d (eval)
  * networks/models/aot.py:107-142               encode / id-bank / decode wiring
  * networks/layers/position.py:50-77            2-D sine positional embedding
  * networks/layers/attention.py:28-81           multi-head attention (explicit + SDPA paths)
  * networks/layers/transformer.py:199-436, 553-692   LSTT block, stack, memory update, eviction
  * networks/layers/basic.py:15-35               GroupNorm + GELU + depth-wise 5x5
  * networks/decoders/fpn.py:36-68               FPN segmentation head
  * networks/engines/aot_engine.py:208-483, 571-725   per-clip engine protocol
  * networks/managers/evaluator.py:330-335, 385-523   per-frame evaluator protocol

It is written functionally over a flat ``{name: tensor}`` weight dict (the
reference's state_dict keys) and plain torch CPU ops in fp32; the dispatch into
``F.scaled_dot_product_attention`` vs. the explicit softmax is kept where the
reference has it so that the oracle is bit-comparable with the reference on this
machine.  Parity pin: tests/test_oracle_golden.py checks this file against
fixtures produced by importing the reference itself (tests/golden/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
W = Dict[str, Tensor]

R50_BLOCKS = (3, 4, 6)        # resnet.py:359-374 with stage 5 dropped (192-193)
R50_STRIDES = (1, 2, 2)       # resnet.py:82-84 (output_stride 16)
BN_EPS = 1e-5                 # normalization.py:11
NUM_HEADS = 8                 # configs/models/default.py:19-20
MAX_OBJ = 10                  # configs/models/default.py:17


# ----------------------------------------------------------------------------
# encoder
# ----------------------------------------------------------------------------
def frozen_bn(x: Tensor, w: W, p: str) -> Tensor:
    """normalization.py:35-43 (no-grad branch)."""
    return F.batch_norm(x, w[p + '.running_mean'], w[p + '.running_var'], w[p + '.weight'],
                        w[p + '.bias'], training=False, eps=BN_EPS)


def bottleneck(x: Tensor, w: W, p: str, stride: int) -> Tensor:
    """resnet.py:48-68; the stride sits on the 3x3 conv (27-35)."""
    out = F.relu(frozen_bn(F.conv2d(x, w[p + '.conv1.weight']), w, p + '.bn1'))
    out = F.relu(frozen_bn(F.conv2d(out, w[p + '.conv2.weight'], stride=stride, padding=1), w, p + '.bn2'))
    out = frozen_bn(F.conv2d(out, w[p + '.conv3.weight']), w, p + '.bn3')
    if (p + '.downsample.0.weight') in w:
        x = frozen_bn(F.conv2d(x, w[p + '.downsample.0.weight'], stride=stride), w, p + '.downsample.1')
    return F.relu(out + x)


def resnet50(img: Tensor, w: W) -> List[Tensor]:
    """resnet.py:178-195 -> [4x, 8x, 16x]."""
    x = F.relu(frozen_bn(F.conv2d(img, w['encoder.conv1.weight'], stride=2, padding=3), w, 'encoder.bn1'))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = []
    for li, (nblk, stride) in enumerate(zip(R50_BLOCKS, R50_STRIDES), start=1):
        for b in range(nblk):
            x = bottleneck(x, w, f'encoder.layer{li}.{b}', stride if b == 0 else 1)
        outs.append(x)
    return outs


# --- Swin-B encoder (cfg 5): encoders/swin/swin_transformer.py, last stage dropped (571) ---------------------
SWIN_DEPTHS, SWIN_HEADS, SWIN_WS = (2, 2, 18), (4, 8, 16), 7       # encoders/swin/build.py:11-22


def _win_part(x: Tensor, ws: int) -> Tensor:
    """swin_transformer.py:65-79: [B,H,W,C] -> [nW*B, ws*ws, C]."""
    B, H, Wd, C = x.shape
    return x.view(B, H // ws, ws, Wd // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def _win_rev(wins: Tensor, ws: int, H: int, Wd: int) -> Tensor:
    """swin_transformer.py:82-97."""
    B = wins.shape[0] // ((H // ws) * (Wd // ws))
    return wins.view(B, H // ws, Wd // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, Wd, -1)


def swin_shift_mask(H: int, Wd: int, ws: int = SWIN_WS) -> Tensor:
    """Attention mask of the shifted windows (swin_transformer.py:432-451): [nW, ws*ws, ws*ws], 0 or -100."""
    sh = ws // 2
    Hp, Wp = -(-H // ws) * ws, -(-Wd // ws) * ws
    img = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
        for wsl in (slice(0, -ws), slice(-ws, -sh), slice(-sh, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = _win_part(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_block(x: Tensor, H: int, Wd: int, w: W, p: str, heads: int, shift: int, mask: Optional[Tensor]) -> Tensor:
    """swin_transformer.py:249-320 (SwinTransformerBlock.forward) with WindowAttention.forward (156-195)."""
    ws = SWIN_WS
    B, L, C = x.shape
    y = F.layer_norm(x, (C,), w[p + '.norm1.weight'], w[p + '.norm1.bias'], 1e-5).view(B, H, Wd, C)
    pr, pb = (ws - Wd % ws) % ws, (ws - H % ws) % ws
    y = F.pad(y, (0, 0, 0, pr, 0, pb))                 # zero tokens AFTER the norm: their q/k/v are the qkv bias
    Hp, Wp = y.shape[1], y.shape[2]
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    xw = _win_part(y, ws)
    B_, N, _ = xw.shape
    qkv = F.linear(xw, w[p + '.attn.qkv.weight'], w[p + '.attn.qkv.bias']).reshape(B_, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // heads) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = w[p + '.attn.relative_position_bias_table'][w[p + '.attn.relative_position_index'].view(-1)].view(N, N, -1)
    attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
    if shift > 0:
        nW = mask.shape[0]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    o = F.linear((attn @ v).transpose(1, 2).reshape(B_, N, C), w[p + '.attn.proj.weight'], w[p + '.attn.proj.bias'])
    y = _win_rev(o, ws, Hp, Wp)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    x = x + y[:, :H, :Wd, :].reshape(B, H * Wd, C)
    h = F.layer_norm(x, (C,), w[p + '.norm2.weight'], w[p + '.norm2.bias'], 1e-5)
    h = F.linear(F.gelu(F.linear(h, w[p + '.mlp.fc1.weight'], w[p + '.mlp.fc1.bias'])), w[p + '.mlp.fc2.weight'], w[p + '.mlp.fc2.bias'])
    return x + h


def swin_patch_merge(x: Tensor, H: int, Wd: int, w: W, p: str) -> Tensor:
    """swin_transformer.py:336-356 (PatchMerging.forward)."""
    B, L, C = x.shape
    x = x.view(B, H, Wd, C)
    if H % 2 or Wd % 2:
        x = F.pad(x, (0, 0, 0, Wd % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), w[p + '.norm.weight'], w[p + '.norm.bias'], 1e-5)
    return F.linear(x, w[p + '.reduction.weight'])


def swin_base(img: Tensor, w: W) -> List[Tensor]:
    """swin_transformer.py:684-716 -> [4x (128 ch), 8x (256), 16x (512)] NCHW."""
    _, _, H, Wd = img.shape
    if Wd % 4:
        img = F.pad(img, (0, 4 - Wd % 4))
    if H % 4:
        img = F.pad(img, (0, 0, 0, 4 - H % 4))
    x = F.conv2d(img, w['encoder.patch_embed.proj.weight'], w['encoder.patch_embed.proj.bias'], stride=4)
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), w['encoder.patch_embed.norm.weight'], w['encoder.patch_embed.norm.bias'], 1e-5)
    outs = []
    for li, (depth, heads) in enumerate(zip(SWIN_DEPTHS, SWIN_HEADS)):
        mask = swin_shift_mask(Wh, Ww)
        for b in range(depth):
            x = swin_block(x, Wh, Ww, w, f'encoder.layers.{li}.blocks.{b}', heads, 0 if b % 2 == 0 else SWIN_WS // 2, mask)
        C = x.shape[-1]
        o = F.layer_norm(x, (C,), w[f'encoder.norm{li}.weight'], w[f'encoder.norm{li}.bias'], 1e-5)
        outs.append(o.view(-1, Wh, Ww, C).permute(0, 3, 1, 2).contiguous())
        if li < len(SWIN_DEPTHS) - 1:
            x = swin_patch_merge(x, Wh, Ww, w, f'encoder.layers.{li}.downsample')
            Wh, Ww = (Wh + 1) // 2, (Ww + 1) // 2
    return outs


def encode_image(img: Tensor, w: W) -> List[Tensor]:
    """aot.py:116-134: [4x, 8x, 16x, proj(16x)]."""
    xs = swin_base(img, w) if 'encoder.patch_embed.proj.weight' in w else resnet50(img, w)
    xs.append(F.conv2d(xs[-1], w['encoder_projector.weight'], w['encoder_projector.bias']))
    return xs


# ----------------------------------------------------------------------------
# embeddings
# ----------------------------------------------------------------------------
def sine_pos_emb(h: int, wd: int, c: int = 256) -> Tensor:
    """position.py:50-77 with normalize=True, scale 2*pi, temperature 1e4 -> [h*w, 1, c]."""
    nf = c // 2
    ys = torch.arange(h, dtype=torch.float32)[:, None].expand(h, wd)
    xs = torch.arange(wd, dtype=torch.float32)[None, :].expand(h, wd)
    eps = 1e-6
    ys = ys / (ys[-1:, :] + eps) * (2 * math.pi)
    xs = xs / (xs[:, -1:] + eps) * (2 * math.pi)
    dim_t = torch.arange(nf, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode='trunc') / nf)
    px = xs[:, :, None] / dim_t
    py = ys[:, :, None] / dim_t
    px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
    py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
    pos = torch.cat((py, px), dim=2)          # [h, w, c], y half first
    return pos.reshape(h * wd, 1, c)


def one_hot_mask(mask: Tensor, cls_num: int = MAX_OBJ) -> Tuple[Tensor, Tensor]:
    """utils/image.py:69-74."""
    if mask.dim() == 3:
        mask = mask.unsqueeze(1)
    idx = torch.arange(0, cls_num + 1).view(1, -1, 1, 1)
    return (mask == idx).float(), (mask == 255).float()


def assign_identity(one_hot: Tensor, ignore: Optional[Tensor], w: W, align_corners: bool = True) -> Tensor:
    """aot_engine.py:208-232 + aot.py:111-114 -> [HW, 1, 256]."""
    if ignore is None:
        ignore = torch.zeros(one_hot.shape[0], 1, one_hot.shape[2], one_hot.shape[3])
    one_hot = one_hot.clone()
    one_hot[:, 0] = one_hot[:, 0] * (ignore == 0).float().squeeze()
    x = torch.cat((one_hot, ignore), 1)
    if align_corners:
        e = F.conv2d(x, w['patch_wise_id_bank.weight'], w['patch_wise_id_bank.bias'], stride=16, padding=8)
    else:
        e = F.conv2d(x, w['patch_wise_id_bank.weight'], w['patch_wise_id_bank.bias'], stride=16, padding=0)
    return e.flatten(2).permute(2, 0, 1)


# ----------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------
def mha(Q: Tensor, K: Tensor, V: Tensor, w: W, p: str, use_linear: bool, explicit: bool,
        heads: int = NUM_HEADS) -> Tuple[Tensor, Optional[Tensor]]:
    """attention.py:28-81.  Q [Lq,B,C], K/V [Lk,B,C] -> ([Lq,B,C], attn [B,h,Lq,Lk] or None)."""
    bs, C = Q.shape[1], Q.shape[2]
    d = C // heads
    if use_linear:
        Q = F.linear(Q, w[p + '.linear_Q.weight'], w[p + '.linear_Q.bias'])
        K = F.linear(K, w[p + '.linear_K.weight'], w[p + '.linear_K.bias'])
        V = F.linear(V, w[p + '.linear_V.weight'], w[p + '.linear_V.bias'])
    if explicit:
        Q = Q / (C / heads) ** 0.5
        q = Q.view(-1, bs, heads, d).permute(1, 2, 0, 3)
        k = K.view(-1, bs, heads, d).permute(1, 2, 3, 0)
        v = V.view(-1, bs, heads, d).permute(1, 2, 0, 3)
        attn = torch.softmax(q @ k, dim=-1)
        out = (attn @ v).permute(2, 0, 1, 3)
    else:
        q = Q.view(-1, bs, heads, d).permute(1, 2, 0, 3)
        k = K.view(-1, bs, heads, d).permute(1, 2, 0, 3)
        v = V.view(-1, bs, heads, d).permute(1, 2, 0, 3)
        out = F.scaled_dot_product_attention(q, k, v, None, 0.0, is_causal=False).permute(2, 0, 1, 3)
        attn = None
    out = out.reshape(-1, bs, C)
    return F.linear(out, w[p + '.projection.weight'], w[p + '.projection.bias']), attn


def temporal_slots(T: int, n_slots: int = 4) -> List[int]:
    """Which of the 4 learned memory PEs each bank entry gets (transformer.py:598-621).

    T <= 4: entry t takes slot t (linear interpolation of T points to T points is the
    identity).  T > 4: the 4 slots are flipped, nearest-resized to T and flipped back.
    ``F.interpolate(mode='nearest')`` picks src = floor(dst * float32(4 / T)).
    """
    if T <= n_slots:
        return list(range(T))
    scale = np.float32(n_slots) / np.float32(T)
    out = []
    for t in range(T):
        src = int(math.floor(np.float32(T - 1 - t) * scale))
        out.append(n_slots - 1 - min(src, n_slots - 1))
    return out


def gn_gelu_dwconv(x: Tensor, size_2d: Tuple[int, int], w: W, p: str) -> Tensor:
    """basic.py:27-35: GroupNorm(32) -> GELU -> depth-wise 5x5, on [HW,B,C]."""
    h, wd = size_2d
    _, bs, c = x.shape
    x = x.view(h, wd, bs, c).permute(2, 3, 0, 1)
    x = F.group_norm(x, 32, w[p + '.gn.weight'], w[p + '.gn.bias'], 1e-5)
    x = F.gelu(x)
    x = F.conv2d(x, w[p + '.conv.weight'], None, padding=2, groups=c)
    return x.reshape(bs, c, h * wd).permute(2, 0, 1)


def ln(x: Tensor, w: W, p: str) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), w[p + '.weight'], w[p + '.bias'], 1e-5)


def lstt_block(x: Tensor, w: W, p: str, long_mem, short_mem, curr_id_emb, self_pos, size_2d,
               temporal: Optional[Tensor], save_attn: bool):
    """transformer.py:553-692 (linear_q=False branch).  Returns (x, memories, record)."""
    # self-attention (565-571) -- always the SDPA path (attention.py:65-74)
    t1 = ln(x, w, p + '.norm1')
    qk = t1 + self_pos
    x = x + mha(qk, qk, t1, w, p + '.self_attn', True, False)[0]

    # long/short-term attention (573-592)
    t2 = ln(x, w, p + '.norm2')
    curr_Q = F.linear(t2, w[p + '.linear_Q.weight'], w[p + '.linear_Q.bias'])
    curr_K, curr_V = curr_Q, t2
    if curr_id_emb is not None:
        global_V = F.linear(curr_V + curr_id_emb, w[p + '.linear_V.weight'], w[p + '.linear_V.bias'])
        local_K, local_V = curr_K, global_V
        global_K, global_V = curr_K[None], global_V[None]
    else:
        global_K, global_V = long_mem
        local_K, local_V = short_mem

    # temporal positional embedding (594-630)
    T, L, bs, C = global_K.shape
    if temporal is not None:
        cur_pe, mem_pe = temporal[0:1], temporal[1:]
        slots = temporal_slots(T, mem_pe.shape[0])
        pe = mem_pe[slots]                                   # [T, C]
        flat_K = (global_K + pe.view(T, 1, 1, C)).flatten(0, 1)
        q_time = curr_Q + cur_pe.view(1, 1, C)
    else:
        flat_K, q_time = global_K.flatten(0, 1), curr_Q
    flat_V = global_V.flatten(0, 1)

    tgt2, attn = mha(q_time, flat_K, flat_V, w, p + '.long_term_attn', False, save_attn)
    record = None
    if save_attn:                                            # 636-643
        a = attn.view(bs, NUM_HEADS, L, T, L).mean(dim=1)[0]  # [L, T, L]
        record = a.sum(dim=2)                                # [L, T]

    k4 = ln(local_K + curr_K, w, p + '.norm4')               # 656-662
    v4 = ln(local_V + curr_V, w, p + '.norm4')
    tgt3, _ = mha(curr_Q, k4, v4, w, p + '.short_term_attn', False, save_attn)

    local_K = F.linear(tgt3, w[p + '.linear_QMem.weight'], w[p + '.linear_QMem.bias'])  # 675
    local_V = tgt3
    if curr_id_emb is not None:                              # 677-678
        local_V = F.linear(local_V + curr_id_emb, w[p + '.linear_VMem.weight'], w[p + '.linear_VMem.bias'])

    x = x + tgt2 + tgt3                                      # 680
    t3 = ln(x, w, p + '.norm3')                              # 683-687
    h = F.linear(t3, w[p + '.linear1.weight'], w[p + '.linear1.bias'])
    h = gn_gelu_dwconv(h, size_2d, w, p + '.activation')
    x = x + F.linear(h, w[p + '.linear2.weight'], w[p + '.linear2.bias'])
    return x, [[curr_K, curr_V], [global_K, global_V], [local_K, local_V]], record


def fpn_decode(inputs: Sequence[Tensor], shortcuts: Sequence[Tensor], w: W, align_corners: bool = True) -> Tensor:
    """decoders/fpn.py:36-68 -> logits [B, 11, H/4, W/4]."""
    def convgn(x, p, k, groups=8):
        x = F.conv2d(x, w[p + '.conv.weight'], w[p + '.conv.bias'], padding=k // 2)
        return F.group_norm(x, groups, w[p + '.gn.weight'], w[p + '.gn.bias'], 1e-5)

    def adapter(x, p):
        return F.conv2d(x, w[p + '.weight'], w[p + '.bias'])

    x = F.relu(convgn(torch.cat(list(inputs), dim=1), 'decoder.conv_in', 1))
    x = F.relu(convgn(adapter(shortcuts[-2], 'decoder.adapter_16x') + x, 'decoder.conv_16x', 3))
    x = F.interpolate(x, size=shortcuts[-3].shape[-2:], mode='bilinear', align_corners=align_corners)
    x = F.relu(convgn(adapter(shortcuts[-3], 'decoder.adapter_8x') + x, 'decoder.conv_8x', 3))
    x = F.interpolate(x, size=shortcuts[-4].shape[-2:], mode='bilinear', align_corners=align_corners)
    x = F.relu(convgn(adapter(shortcuts[-4], 'decoder.adapter_4x') + x, 'decoder.conv_4x', 3))
    return F.conv2d(x, w['decoder.conv_out.weight'], w['decoder.conv_out.bias'])


# ----------------------------------------------------------------------------
# memory eviction policy
# ----------------------------------------------------------------------------
class EvictionState:
    """EMA scores and visit counts kept across evictions (transformer.py:442-443)."""

    def __init__(self):
        self.ema: Dict[int, float] = {}
        self.visits: Dict[int, int] = {}


def choose_eviction(attn_mass: Tensor, fg_prob: Tensor, indexes: List[int], st: EvictionState) -> int:
    """transformer.py:338-411 (eval branch): which bank entry to drop.

    attn_mass [HW, T'] is layer 0's per-memory-frame probability mass of the frame just
    propagated (T' = bank size before this frame was appended, so ``indexes`` has T'+1
    entries); fg_prob [HW] = 1 - P(background).
    """
    a = (attn_mass * fg_prob.reshape(-1, 1)).sum(dim=0)
    a = (a / a.sum()).clone()
    cur = {indexes[i]: a[i] for i in range(a.shape[0])}
    cur = {k: (0.2 * st.ema[k] + 0.8 * v) if k in st.ema else v for k, v in cur.items()}
    st.ema = cur
    for i in range(a.shape[0]):
        a[i] = cur[indexes[i]]
    visits = {k: 1 + st.visits.get(k, 0) for k in indexes}
    st.visits = visits
    n = torch.tensor([float(visits[k]) for k in indexes[:-1]])
    n[0] = len(n)
    a = a + 1.5 * torch.sqrt(torch.log(n.sum()) / (n + 8))
    rest = a[1:]
    return int(torch.argmin(rest).item()) + 1 if rest.shape[0] > 0 else 1


# ----------------------------------------------------------------------------
# engine
# ----------------------------------------------------------------------------
class OracleEngine:
    """One AOTEngine (<= 10 objects) driven the way AOTInferEngine drives it.

    aot_engine.py:241-325 (reference frame), 398-465 (propagate), 327-369 (memory
    update + restriction), 571-725 (infer-engine wrapper; obj_nums is forced to
    [max_obj_num], 697).
    """

    def __init__(self, weights: W, former_len: int = 1, latter_len: int = 7, long_term_mem_gap: int = 5,
                 num_lstt: int = 3, align_corners: bool = True):
        self.w = weights
        self.former, self.latter = former_len, latter_len
        self.long_term_mem_gap = long_term_mem_gap
        self.L = num_lstt
        self.align_corners = align_corners
        self.restart_engine()

    def restart_engine(self):
        self.frame_step = 0
        self.last_mem_step = -1
        self.pos_emb = None
        self.long_mem = None
        self.short_mem = None
        self.long_memories_indexes: List[int] = []
        self.evict = EvictionState()
        self.drop_trace: List[int] = []
        self.input_size_2d = None

    @property
    def temporal(self):
        return torch.cat((self.w['cur_pos_emb'], self.w['mem_pos_emb']), dim=0)

    def _lstt(self, xs, id_emb, save_attn):
        """transformer.py:199-267."""
        x = xs[-1].flatten(2).permute(2, 0, 1).contiguous()   # bchw_2_lbc, utils/tensor.py:3-6
        outs, mems, rec0 = [], [], None
        for i in range(self.L):
            x, m, rec = lstt_block(x, self.w, f'LSTT.layers.{i}',
                                   self.long_mem[i] if self.long_mem is not None else None,
                                   self.short_mem[i] if self.short_mem is not None else None,
                                   id_emb, self.pos_emb, self.enc_size_2d, self.temporal, save_attn)
            outs.append(x)
            mems.append(m)
            if i == 0:
                rec0 = rec
        # decoder_norms: final norm on the last, intermediate norms on the rest (248-259)
        outs = [ln(o, self.w, f'LSTT.decoder_norms.{i}') for i, o in enumerate(outs)]
        self.curr_mem = [m[0] for m in mems]
        self.lstt_long = [m[1] for m in mems]
        self.lstt_short = [m[2] for m in mems]
        if save_attn:
            self.record_attn_weight = rec0
        return outs

    def _decode(self, xs, lstt_outs, output_size):
        n, c, h, wd = xs[-1].shape
        ins = [xs[-1]] + [o.view(h, wd, n, c).permute(2, 3, 0, 1) for o in lstt_outs]
        logits = fpn_decode(ins, xs, self.w, self.align_corners)
        # obj_nums == [10] (aot_engine.py:697) so no channel is masked at 451-453
        self.pred_id_logits = logits
        if output_size is not None:
            logits = F.interpolate(logits, size=output_size, mode='bilinear', align_corners=self.align_corners)
        return logits

    def _assign_identity(self, oh, ign):
        return assign_identity(oh, ign, self.w, self.align_corners)

    def add_reference_frame(self, img: Tensor, mask: Tensor, frame_step: int = 0):
        xs = encode_image(img, self.w)
        if self.input_size_2d is None:
            self.input_size_2d = tuple(img.shape[2:])
            self.enc_size_2d = tuple(xs[-1].shape[2:])
        if self.pos_emb is None:
            self.pos_emb = sine_pos_emb(*self.enc_size_2d)
        oh, ign = one_hot_mask(mask)
        id_emb = self._assign_identity(oh, None)                        # ignore mask not forwarded (305)
        outs = self._lstt(xs, id_emb, False)
        self.last_mem_step = frame_step
        self.long_mem = [list(m) for m in self.lstt_long]               # init_memory, transformer.py:438-443
        self.short_mem = [list(m) for m in self.lstt_short]
        self.evict = EvictionState()
        self.long_memories_indexes.append(self.frame_step)              # quirk: self.frame_step (323)
        return self._decode(xs, outs, None)

    def match_propogate_one_frame(self, img: Tensor, output_size=None) -> Tensor:
        self.frame_step += 1
        xs = encode_image(img, self.w)
        outs = self._lstt(xs, None, True)
        return self._decode(xs, outs, output_size)

    def update_memory(self, curr_mask: Tensor):
        """aot_engine.py:327-369 + transformer.py:269-322."""
        if curr_mask.dim() == 3 or curr_mask.shape[0] == 1 and curr_mask.shape[1] == 1:
            oh, ign = one_hot_mask(curr_mask)
        else:
            oh, ign = curr_mask, None
        id_emb = self._assign_identity(oh, ign)
        update_long = (self.frame_step - self.last_mem_step) >= self.long_term_mem_gap
        if update_long:
            self.last_mem_step = self.frame_step
        self._update_memories(id_emb, update_long)

    def _update_memories(self, id_emb, update_long):
        for i in range(self.L):
            p = f'LSTT.layers.{i}'
            self.curr_mem[i][1] = F.linear(self.curr_mem[i][1] + id_emb, self.w[p + '.linear_V.weight'], self.w[p + '.linear_V.bias'])
            self.lstt_short[i][1] = F.linear(self.lstt_short[i][1] + id_emb, self.w[p + '.linear_VMem.weight'], self.w[p + '.linear_VMem.bias'])
        self.short_mem = [[k, v] for k, v in self.lstt_short]
        if not update_long:
            return
        for i in range(self.L):
            self.long_mem[i] = [torch.cat([old, new[None]], dim=0) for old, new in zip(self.long_mem[i], self.curr_mem[i])]
        self.long_memories_indexes.append(self.frame_step)
        logits = F.interpolate(self.pred_id_logits, size=self.enc_size_2d, mode='bilinear', align_corners=True)
        fg = 1 - torch.softmax(logits, dim=1)[:, 0:1]
        if self.long_mem[0][0].shape[0] <= self.former + self.latter:
            return
        drop = choose_eviction(self.record_attn_weight, fg.flatten(), self.long_memories_indexes, self.evict)
        self.drop_trace.append(drop)
        for i in range(self.L):
            self.long_mem[i] = [torch.cat([m[:drop], m[drop + 1:]], dim=0) for m in self.long_mem[i]]
        del self.long_memories_indexes[drop]


class OracleInferEngine:
    """AOTInferEngine (aot_engine.py:571-725): one OracleEngine per 10 objects, label masks split per engine
    (604-618), logits merged by soft aggregation (650-673)."""

    def __init__(self, weights: W, former_len=1, latter_len=7, long_term_mem_gap=5, max_aot_obj_num=MAX_OBJ):
        self.args = (weights, former_len, latter_len, long_term_mem_gap)
        self.max_aot_obj_num = max_aot_obj_num
        self.engines: List[OracleEngine] = []

    def separate_mask(self, mask):
        if len(self.engines) == 1:
            return [mask]
        out = []
        for idx in range(len(self.engines)):
            start_id = idx * self.max_aot_obj_num + 1
            end_id = (idx + 1) * self.max_aot_obj_num
            fg = ((mask >= start_id) & (mask <= end_id)).float()
            out.append((fg * mask - start_id + 1) * fg)
        return out

    def soft_logit_aggregation(self, all_logits):
        if len(all_logits) == 1:
            return all_logits[0]
        fg, bg = [], []
        for logit in all_logits:
            prob = torch.softmax(logit, dim=1)
            bg.append(prob[:, 0:1])
            fg.append(prob[:, 1:1 + self.max_aot_obj_num])
        bg_prob = torch.prod(torch.cat(bg, dim=1), dim=1, keepdim=True)
        return torch.logit(torch.cat([bg_prob] + fg, dim=1).clamp(1e-5, 1 - 1e-5))

    def add_reference_frame(self, img, mask, obj_nums, frame_step=0):
        n = max(int(np.ceil(obj_nums / self.max_aot_obj_num)), 1)
        while len(self.engines) < n:
            self.engines.append(OracleEngine(*self.args))
        for e, m in zip(self.engines, self.separate_mask(mask)):
            e.add_reference_frame(img, m, frame_step)
        self.input_size_2d = self.engines[0].input_size_2d

    def match_propogate_one_frame(self, img, output_size=None):
        return self.soft_logit_aggregation([e.match_propogate_one_frame(img, output_size) for e in self.engines])

    def update_memory(self, mask):
        for e, m in zip(self.engines, self.separate_mask(mask)):
            e.update_memory(m)

    @property
    def long_memories_indexes(self):
        return self.engines[0].long_memories_indexes


def run_clip(engine: OracleEngine, frames: Sequence[Tensor], first_mask: Tensor, output_size: Tuple[int, int],
             gap: Optional[int] = None):
    """The evaluator's per-frame protocol (evaluator.py:330-335, 385-441, 509-523) for one clip.

    frames: list of [1,3,H,W] normalised images at network size; first_mask [1,1,H,W] int
    labels at network size.  Returns (list of label maps [Ho,Wo] int64, list of logits).
    """
    n = len(frames)
    engine.long_term_mem_gap = gap if gap is not None else max(int(round(n / 30)), 5)
    engine.add_reference_frame(frames[0], first_mask, frame_step=0)
    labels, all_logits = [], []
    for f in frames[1:]:
        logit = engine.match_propogate_one_frame(f, output_size=output_size)
        prob = torch.softmax(logit, dim=1)
        label = torch.argmax(prob, dim=1, keepdim=True).float()
        engine.update_memory(F.interpolate(label, size=engine.input_size_2d, mode='nearest'))
        labels.append(label[0, 0].long())
        all_logits.append(logit)
    return labels, all_logits


def tta_merge(logits, flips):
    """managers/evaluator.py:427-441: each augmentation's logits [1, nc, H, W] are un-flipped (utils/image.py:109-113, dim 3),
    soft-maxed over the classes, the probabilities averaged over the augmentations and arg-maxed.  Returns (mean probability
    [1, nc, H, W], label [1, 1, H, W] fp32).  Pinned by tests/golden/tta.npz (the reference's own flip_tensor + mean + argmax)."""
    ps = [torch.softmax(lg.flip(3) if fl else lg, dim=1) for lg, fl in zip(logits, flips)]
    prob = torch.mean(torch.cat(ps, 0), dim=0, keepdim=True)
    return prob, torch.argmax(prob, dim=1, keepdim=True).float()


def evaluate_sequence(weights: W, frames, labels: Dict[int, Tensor], out_hw, former=1, latter=7, flip=False):
    """managers/evaluator.py:330-523 for one sequence (gap heuristic, flip / multi-scale TTA with probability averaging over
    one engine per (scale, flip) pair, new-object reference frames, memory update), on OracleInferEngine(s).  frames: one
    [n,3,H,W] tensor or a list of them (one per test scale).  Returns (label maps, mean probabilities) per frame > 0."""
    per_scale = list(frames) if isinstance(frames, (list, tuple)) else [frames]
    n = per_scale[0].shape[0]
    gap = max(int(round(n / 30)), 5)
    augs = [(si, fl) for si in range(len(per_scale)) for fl in ([False, True] if flip else [False])]
    engines = [OracleInferEngine(weights, former, latter, gap) for _ in augs]

    def frame(a, t):
        si, fl = augs[a]
        img = per_scale[si][t:t + 1]
        return img.flip(3) if fl else img

    for a, (e, (si, fl)) in enumerate(zip(engines, augs)):
        lab = F.interpolate(labels[0], size=tuple(per_scale[si].shape[2:]), mode='nearest')
        e.add_reference_frame(frame(a, 0), lab.flip(3) if fl else lab, int(labels[0].max()), 0)
    outs, probs = [], []
    for t in range(1, n):
        lgs = [e.match_propogate_one_frame(frame(a, t), out_hw) for a, e in enumerate(engines)]
        prob, label = tta_merge(lgs, [fl for _, fl in augs])
        if t in labels:
            keep = (labels[t] == 0).float()
            label = label * keep + labels[t] * (1 - keep)
            for a, (e, (si, fl)) in enumerate(zip(engines, augs)):
                lab = F.interpolate(label.flip(3) if fl else label, size=e.input_size_2d, mode='nearest')
                e.add_reference_frame(frame(a, t), lab, int(label.max()), t)
        else:
            for e, (si, fl) in zip(engines, augs):
                e.update_memory(F.interpolate(label.flip(3) if fl else label, size=e.input_size_2d, mode='nearest'))
        outs.append(label[0, 0].to(torch.uint8))
        probs.append(prob)
    return outs, probs


def ingest_rgb8(rgb: np.ndarray, hd: int, wd: int) -> np.ndarray:
    """Frame ingest of the reference data path: float32 image -> cv2.resize(INTER_CUBIC) -> /255, ImageNet mean/std
    (dataloaders/eval_datasets.py:57-64; video_transforms.py:648-652, 676-680).  cv2 is absent here, so the resize is a
    restatement of OpenCV's published bicubic (a = -0.75, fx = (dx + 0.5) * scale - 0.5, taps clamped to the border):
    PARITY UNPINNED for this function.  rgb: uint8 [Hs, Ws, 3] -> float32 [3, hd, wd]."""
    hs, ws = rgb.shape[:2]
    img = rgb.astype(np.float32)

    def taps(n_dst, n_src):
        f = (np.arange(n_dst, dtype=np.float32) + 0.5) * np.float32(n_src / n_dst) - 0.5
        s0 = np.floor(f).astype(np.int64)
        t = (f - s0).astype(np.float32)
        A = np.float32(-0.75)
        w0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
        w1 = ((A + 2) * t - (A + 3)) * t * t + 1
        w2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
        w3 = 1 - w0 - w1 - w2
        idx = np.clip(s0[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, np.stack([w0, w1, w2, w3], 1).astype(np.float32)

    if (hs, ws) != (hd, wd):
        iy, wy = taps(hd, hs)
        ix, wx = taps(wd, ws)
        rows = (img[:, ix, :] * wx[None, :, :, None]).sum(2)          # [hs, wd, 3]
        img = (rows[iy, :, :] * wy[:, :, None, None]).sum(1)          # [hd, wd, 3]
    img = (img / np.float32(255.) - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    return img.transpose(2, 0, 1).astype(np.float32)


def db_eval_iou(annotation: np.ndarray, segmentation: np.ndarray, void_pixels: Optional[np.ndarray] = None) -> float:
    """evaluation/source/metrics.py:6-37 (single frame): Jaccard index of two binary maps, void pixels excluded from both
    counts, 1 when the union is empty."""
    a = annotation.astype(bool)
    s = segmentation.astype(bool)
    keep = np.ones_like(a) if void_pixels is None else np.logical_not(void_pixels.astype(bool))
    union = np.sum((a | s) & keep)
    return 1.0 if union == 0 else float(np.sum((a & s) & keep)) / float(union)
