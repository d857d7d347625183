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

BN_EPS = 1e-5
R50_BLOCKS = (3, 4, 6)
R50_STRIDES = (1, 2, 2)


def _conv_w(w: torch.Tensor, cin_pad: int = 0) -> torch.Tensor:
    cout, cin, kh, kw = w.shape
    w = w.permute(0, 2, 3, 1)
    if cin_pad and cin_pad > cin:
        w = torch.nn.functional.pad(w, (0, cin_pad - cin))
    return w.contiguous().to(torch.bfloat16)


def _fold_bn(sd, conv_key: str, bn_prefix: str, cin_pad: int = 0):
    w = sd[conv_key].float()
    scale = sd[bn_prefix + '.weight'].float() * (sd[bn_prefix + '.running_var'].float() + BN_EPS).rsqrt()
    bias = sd[bn_prefix + '.bias'].float() - sd[bn_prefix + '.running_mean'].float() * scale
    return _conv_w(w * scale.view(-1, 1, 1, 1), cin_pad), bias.contiguous()


def pack_state_dict(sd: Dict[str, torch.Tensor], device, num_lstt: int = 3) -> Dict[str, torch.Tensor]:
    P: Dict[str, torch.Tensor] = {}

    def put(name, t):
        P[name] = t.to(device).contiguous()

    w, b = _fold_bn(sd, 'encoder.conv1.weight', 'encoder.bn1', cin_pad=8)
    put('stem.w', w); put('stem.b', b)
    for li, nblk in enumerate(R50_BLOCKS, start=1):
        for bi in range(nblk):
            p = f'encoder.layer{li}.{bi}'
            for j in (1, 2, 3):
                w, b = _fold_bn(sd, f'{p}.conv{j}.weight', f'{p}.bn{j}')
                put(f'{p}.conv{j}.w', w); put(f'{p}.conv{j}.b', b)
            if f'{p}.downsample.0.weight' in sd:
                w, b = _fold_bn(sd, f'{p}.downsample.0.weight', f'{p}.downsample.1')
                put(f'{p}.ds.w', w); put(f'{p}.ds.b', b)
    put('proj.w', _conv_w(sd['encoder_projector.weight'].float())); put('proj.b', sd['encoder_projector.bias'].float())

    def lin(dst, src):
        put(dst + '.w', sd[src + '.weight'].float().to(torch.bfloat16)); put(dst + '.b', sd[src + '.bias'].float())

    def norm(dst, src):
        put(dst + '.g', sd[src + '.weight'].float()); put(dst + '.b', sd[src + '.bias'].float())

    for i in range(num_lstt):
        s, d = f'LSTT.layers.{i}', f'l{i}'
        norm(d + '.ln1', s + '.norm1')
        put(d + '.self_qk.w', torch.cat([sd[s + '.self_attn.linear_Q.weight'], sd[s + '.self_attn.linear_K.weight']], 0).float().to(torch.bfloat16))
        put(d + '.self_qk.b', torch.cat([sd[s + '.self_attn.linear_Q.bias'], sd[s + '.self_attn.linear_K.bias']], 0).float())
        # Q | K | V in one GEMM: q = k = LN1(x) + pos and v = LN1(x) differ only by pos @ Wqk^T, a per-clip constant
        put(d + '.self_qkv.w', torch.cat([sd[s + '.self_attn.linear_Q.weight'], sd[s + '.self_attn.linear_K.weight'],
                                          sd[s + '.self_attn.linear_V.weight']], 0).float().to(torch.bfloat16))
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
