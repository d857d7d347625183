"""state_dict (reference key names, fp32) -> device tensors in the layouts the HIP kernels read.

One-off work at model load: FrozenBatchNorm2d is folded into the preceding conv
(layers/normalization.py:35-43: y = (x - mean) / sqrt(var + eps) * weight + bias), conv
weights go to [Cout, KH, KW, Cin] bf16 with Cin zero-padded to a multiple of 8, the
self-attention Q and K projections are concatenated (they share their input,
layers/transformer.py:567-569), and the depth-wise 5x5 weight is transposed to [25, C].
"""
from __future__ import annotations

from typing import Dict

import torch

import threading

BN_EPS = 1e-5
_DT = torch.bfloat16          # element type of the pack in progress (set by pack_state_dict, under _PACK_LOCK)
_PACK_LOCK = threading.Lock()
R50_BLOCKS = (3, 4, 6)
R50_STRIDES = (1, 2, 2)


def _conv_w(w: torch.Tensor, cin_pad: int = 0) -> torch.Tensor:
    cout, cin, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if cin_pad and cin_pad > cin:
        w = torch.nn.functional.pad(w, (0, cin_pad - cin))
    return w.contiguous().to(_DT)


def pack_frag(w: torch.Tensor, waves: int = 0) -> torch.Tensor:
    """[N, K] 16-bit weight (nn.Linear layout, N % 256 == 0, K % 32 == 0) -> the fragment order the LSTT chain kernels stream
    (include/rmem.h, rmem_lstt_chain_*): [N / 256][NW waves][K / 32][16 / NW column tiles][64 lanes][8], element (n, k) with
    n = 256 nb + (256 / NW) wave + 16 j + (lane & 15), k = 32 kc + 8 (lane >> 4) + e -- one MFMA B fragment per KiB.
    waves = 0: what the loaded library was built for (rmem_lstt_chain_waves)."""
    if not waves:
        from . import _lib
        waves = _lib.lib().rmem_lstt_chain_waves()
    N, K = w.shape
    assert N % 256 == 0 and K % 32 == 0 and waves in (4, 8), (N, K, waves)
    v = w.reshape(N // 256, waves, 16 // waves, 16, K // 32, 4, 8)          # nb, wave, j, fr, kc, fc, e
    return v.permute(0, 1, 4, 2, 5, 3, 6).contiguous().reshape(-1)            # nb, wave, kc, j, fc, fr, e  (lane = 16 fc + fr)


def _fold_bn(sd, conv_key: str, bn_prefix: str, cin_pad: int = 0):
    w = sd[conv_key].float()
    scale = sd[bn_prefix + '.weight'].float() * (sd[bn_prefix + '.running_var'].float() + BN_EPS).rsqrt()
    bias = sd[bn_prefix + '.bias'].float() - sd[bn_prefix + '.running_mean'].float() * scale
    return _conv_w(w * scale.view(-1, 1, 1, 1), cin_pad), bias.contiguous()


SWIN_DEPTHS, SWIN_HEADS, SWIN_WS = (2, 2, 18), (4, 8, 16), 7     # encoders/swin/build.py:11-22, last stage dropped
LOG2E = 1.4426950408889634


def swin_window_masks(ws: int = SWIN_WS) -> torch.Tensor:
    """Shifted-window attention masks by window type [4, 49, 49] (0 / -100), type = 2 * (last window row) + (last window
    column).  Restates encoders/swin/swin_transformer.py:432-451: inside the last window row (column) the first
    ws - shift rows (columns) and the remaining shift rows (columns) come from different image regions after the roll."""
    sh = ws // 2
    py, px = torch.arange(ws * ws) // ws, torch.arange(ws * ws) % ws
    out = torch.zeros(4, ws * ws, ws * ws)
    for t in range(4):
        rr = (py >= ws - sh).long() if t & 2 else torch.zeros_like(py)
        cr = (px >= ws - sh).long() if t & 1 else torch.zeros_like(px)
        reg = rr * 2 + cr
        out[t] = (reg[:, None] != reg[None, :]).float() * -100.0
    return out


def _pack_swin(sd, put):
    put('pe.w', _conv_w(sd['encoder.patch_embed.proj.weight'].float(), cin_pad=8)); put('pe.b', sd['encoder.patch_embed.proj.bias'].float())
    put('pe.ln.g', sd['encoder.patch_embed.norm.weight'].float()); put('pe.ln.b', sd['encoder.patch_embed.norm.bias'].float())
    masks = swin_window_masks()
    for li, (depth, heads) in enumerate(zip(SWIN_DEPTHS, SWIN_HEADS)):
        for b in range(depth):
            s, d = f'encoder.layers.{li}.blocks.{b}', f'sw{li}.{b}'
            for nm in ('norm1', 'norm2'):
                put(f'{d}.{nm}.g', sd[f'{s}.{nm}.weight'].float()); put(f'{d}.{nm}.b', sd[f'{s}.{nm}.bias'].float())
            for dst, src in (('qkv', 'attn.qkv'), ('proj', 'attn.proj'), ('fc1', 'mlp.fc1'), ('fc2', 'mlp.fc2')):
                put(f'{d}.{dst}.w', sd[f'{s}.{src}.weight'].float().to(_DT)); put(f'{d}.{dst}.b', sd[f'{s}.{src}.bias'].float())
            tbl = sd[f'{s}.attn.relative_position_bias_table'].float()[sd[f'{s}.attn.relative_position_index'].view(-1)]
            bias = tbl.view(49, 49, heads).permute(2, 0, 1)                                   # [heads, q, k]
            put(f'{d}.table', (bias[None] + masks[:, None].to(bias.device)) * LOG2E)           # [4, heads, 49, 49]
        if li < len(SWIN_DEPTHS) - 1:
            s = f'encoder.layers.{li}.downsample'
            put(f'sw{li}.merge.w', sd[s + '.reduction.weight'].float().to(_DT))
            put(f'sw{li}.merge.g', sd[s + '.norm.weight'].float()); put(f'sw{li}.merge.b', sd[s + '.norm.bias'].float())
        put(f'sw.norm{li}.g', sd[f'encoder.norm{li}.weight'].float()); put(f'sw.norm{li}.b', sd[f'encoder.norm{li}.bias'].float())


def pack_deaot_self(sd, p: str):
    """GatedPropagation with use_linear (layers/attention.py:151-173) as ONE GEMM over x = [tgt | tgt_id] (512):
    rows [linear_QK (128) | V = [V1 (x[:256]) | V2 (x[256:])] (1024) | U likewise (1024)]; the two halves of V and U read
    disjoint input halves, so their weight is block diagonal.  Returns (bf16 [2176, 512], fp32 [2176])."""
    f = lambda k: sd[p + k].float()   # noqa: E731
    z = torch.zeros(512, 256, device=sd[p + '.linear_QK.weight'].device)
    blk = lambda a, b: torch.cat([torch.cat([f(a + '.weight'), z], 1), torch.cat([z, f(b + '.weight')], 1)], 0)   # noqa: E731
    W = torch.cat([f('.linear_QK.weight'), blk('.linear_V1', '.linear_V2'), blk('.linear_U1', '.linear_U2')], 0)
    b = torch.cat([f('.linear_QK.bias'), f('.linear_V1.bias'), f('.linear_V2.bias'), f('.linear_U1.bias'), f('.linear_U2.bias')], 0)
    return W.to(_DT), b


def _pack_deaot_gpm(sd, put, lin, norm, num_lstt):
    """DualBranchGPM (layers/transformer.py:700-763, 1011-1082): per layer the fused [Q | V | U] GEMM (linear_QV + linear_U),
    the ID branch linears, the relative embedding, three depth-wise kernels + projections, the fused self-attention GEMM."""
    for i in range(num_lstt):
        s, d = f'LSTT.layers.{i}', f'g{i}'
        norm(d + '.ln1', s + '.norm1')
        put(d + '.qvu.w', torch.cat([sd[s + '.linear_QV.weight'], sd[s + '.linear_U.weight']], 0).float().to(_DT))
        put(d + '.qvu.b', torch.cat([sd[s + '.linear_QV.bias'], sd[s + '.linear_U.bias']], 0).float())
        lin(d + '.idv', s + '.linear_ID_V')
        if i > 0:
            norm(d + '.idn1', s + '.id_norm1')
            lin(d + '.idu', s + '.linear_ID_U')
        put(d + '.rel.w', sd[s + '.short_term_attn.relative_emb_k.weight'].float().reshape(225, 128).to(_DT))
        put(d + '.rel.b', sd[s + '.short_term_attn.relative_emb_k.bias'].float())
        for nm, src in (('long', 'long_term_attn'), ('short', 'short_term_attn'), ('self', 'self_attn')):
            put(f'{d}.{nm}_dw.w', sd[f'{s}.{src}.dw_conv.conv.weight'].float().view(-1, 25).t())
            lin(f'{d}.{nm}_proj', f'{s}.{src}.projection')
        norm(d + '.ln2', s + '.norm2')
        norm(d + '.idn2', s + '.id_norm2')
        W, b = pack_deaot_self(sd, s + '.self_attn')
        put(d + '.self.w', W); put(d + '.self.b', b)
    norm('dec_gn', 'LSTT.decoder_norms.0.gn')
    norm('idnorm', 'id_norm')


def pack_state_dict(sd: Dict[str, torch.Tensor], device, num_lstt: int = 3, dtype=torch.bfloat16) -> Dict[str, torch.Tensor]:
    """dtype: the 16-bit element type of weights / activations / bank (torch.bfloat16, or torch.float16 for the *_f16 entry
    points); weights are rounded from fp32 directly to it."""
    global _DT
    if dtype not in (torch.bfloat16, torch.float16):
        raise ValueError(f'pack_state_dict: dtype must be torch.bfloat16 or torch.float16, got {dtype}')
    with _PACK_LOCK:                  # the helpers below read the element type from the module: one pack at a time
        _DT = dtype
        try:
            return _pack_state_dict(sd, device, num_lstt)
        finally:
            _DT = torch.bfloat16      # the helpers of this module pack bfloat16 unless told otherwise


def _pack_state_dict(sd: Dict[str, torch.Tensor], device, num_lstt: int) -> Dict[str, torch.Tensor]:
    P: Dict[str, torch.Tensor] = {}

    def put(name, t):
        P[name] = t.to(device).contiguous()

    swin = 'encoder.patch_embed.proj.weight' in sd
    if swin:
        _pack_swin(sd, put)
    else:
        w, b = _fold_bn(sd, 'encoder.conv1.weight', 'encoder.bn1', cin_pad=8)
        put('stem.w', w); put('stem.b', b)
        w4 = torch.zeros(w.shape[0], 8, 8, 4, dtype=w.dtype)         # rmem_stem7x7s2: [64][ky 8][kx 8][c 4], zero tail (K = 256)
        w4[:, :7, :7, :3] = w.view(w.shape[0], 7, 7, 8)[..., :3]
        put('stem.w4', w4)
    for li, nblk in enumerate(() if swin else R50_BLOCKS, start=1):
        for bi in range(nblk):
            p = f'encoder.layer{li}.{bi}'
            for j in (1, 2, 3):
                w, b = _fold_bn(sd, f'{p}.conv{j}.weight', f'{p}.bn{j}')
                put(f'{p}.conv{j}.w', w); put(f'{p}.conv{j}.b', b)
            if f'{p}.downsample.0.weight' in sd:
                w, b = _fold_bn(sd, f'{p}.downsample.0.weight', f'{p}.downsample.1')
                put(f'{p}.ds.w', w); put(f'{p}.ds.b', b)
                # conv3 and the shortcut as one GEMM over [conv2 output | block input] (ops.conv1x1_dual)
                w3, b3 = P[f'{p}.conv3.w'], P[f'{p}.conv3.b']
                put(f'{p}.c3ds.w', torch.cat([w3.reshape(w3.shape[0], -1), P[f'{p}.ds.w'].reshape(w3.shape[0], -1)], 1))
                put(f'{p}.c3ds.b', b3 + P[f'{p}.ds.b'])
    put('proj.w', _conv_w(sd['encoder_projector.weight'].float())); put('proj.b', sd['encoder_projector.bias'].float())

    def lin(dst, src):
        put(dst + '.w', sd[src + '.weight'].float().to(_DT)); put(dst + '.b', sd[src + '.bias'].float())

    def norm(dst, src):
        put(dst + '.g', sd[src + '.weight'].float()); put(dst + '.b', sd[src + '.bias'].float())

    deaot = 'LSTT.layers.0.linear_QV.weight' in sd
    if deaot:
        _pack_deaot_gpm(sd, put, lin, norm, num_lstt)
    for i in range(0 if deaot else num_lstt):
        s, d = f'LSTT.layers.{i}', f'l{i}'
        norm(d + '.ln1', s + '.norm1')
        put(d + '.self_qk.w', torch.cat([sd[s + '.self_attn.linear_Q.weight'], sd[s + '.self_attn.linear_K.weight']], 0).float().to(_DT))
        put(d + '.self_qk.b', torch.cat([sd[s + '.self_attn.linear_Q.bias'], sd[s + '.self_attn.linear_K.bias']], 0).float())
        # Q | K | V in one GEMM: q = k = LN1(x) + pos and v = LN1(x) differ only by pos @ Wqk^T, a per-clip constant
        put(d + '.self_qkv.w', torch.cat([sd[s + '.self_attn.linear_Q.weight'], sd[s + '.self_attn.linear_K.weight'],
                                          sd[s + '.self_attn.linear_V.weight']], 0).float().to(_DT))
        put(d + '.self_qkv.b', torch.cat([sd[s + '.self_attn.linear_Q.bias'], sd[s + '.self_attn.linear_K.bias'],
                                          sd[s + '.self_attn.linear_V.bias']], 0).float())
        lin(d + '.self_proj', s + '.self_attn.projection')
        norm(d + '.ln2', s + '.norm2')
        for nm in ('linear_Q', 'linear_V', 'linear_QMem', 'linear_VMem', 'linear1', 'linear2'):
            lin(f'{d}.{nm}', f'{s}.{nm}')
        norm(d + '.ln4', s + '.norm4')
        lin(d + '.long_proj', s + '.long_term_attn.projection')
        lin(d + '.short_proj', s + '.short_term_attn.projection')
        norm(d + '.ln3', s + '.norm3')
        for nm in ('self_proj', 'linear_Q', 'long_proj', 'short_proj', 'linear1', 'linear2', 'self_qkv'):
            put(f'{d}.{nm}.wf', pack_frag(P[f'{d}.{nm}.w']))     # fragment order for the row-chain kernels
        norm(d + '.gn', s + '.activation.gn')
        put(d + '.dw.w', sd[s + '.activation.conv.weight'].float().view(-1, 25).t())
        norm(f'dec_norm{i}', f'LSTT.decoder_norms.{i}')

    for nm in ('conv_in', 'conv_16x', 'conv_8x', 'conv_4x'):
        put(f'dec.{nm}.w', _conv_w(sd[f'decoder.{nm}.conv.weight'].float())); put(f'dec.{nm}.b', sd[f'decoder.{nm}.conv.bias'].float())
        norm(f'dec.{nm}.gn', f'decoder.{nm}.gn')
    for nm in ('adapter_16x', 'adapter_8x', 'adapter_4x', 'conv_out'):
        put(f'dec.{nm}.w', _conv_w(sd[f'decoder.{nm}.weight'].float())); put(f'dec.{nm}.b', sd[f'decoder.{nm}.bias'].float())

    put('idbank.w', _conv_w(sd['patch_wise_id_bank.weight'].float(), cin_pad=16)); put('idbank.b', sd['patch_wise_id_bank.bias'].float())
    put('pe_cur', sd['cur_pos_emb'].float().view(-1)); put('pe_mem', sd['mem_pos_emb'].float())
    return P
