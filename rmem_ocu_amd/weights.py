"""Deterministic synthetic weights for the R50-AOTL(+RMem) inference path.

No trained checkpoint is available offline (reference README.md:57-61 points at
Google-Drive links), so parity fixtures, tests and bench.py all use this
generator.  It produces one fp32 tensor for every key of the reference
``state_dict`` contract (362 tensors; names follow
aot_plus/networks/models/aot.py:12-105, layers/transformer.py:466-545,
encoders/resnet.py:10-196, decoders/fpn.py:7-34) so the same dict loads into the
reference (``load_state_dict``), into the CPU oracle and into the HIP engine.

The scales are chosen so that every stage output is O(1) (the reference's
default init makes the encoder projector output ~20x larger than the LSTT
outputs, which hides the memory-read path from the logits).
Each tensor is drawn from its own PCG64 stream keyed by (seed, crc32(name)), so
adding or re-ordering keys never changes another tensor.
"""
from __future__ import annotations

import math
import os
import zlib
from collections import OrderedDict

import numpy as np
import torch

# ResNet-50 with stage 5 dropped: resnet.py:359-374 (layers [3,4,6,(3)]) and 192-193.
_R50_STAGES = ((1, 64, 3, 64), (2, 128, 4, 256), (3, 256, 6, 512))  # (idx, planes, blocks, inplanes)


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std):
    return torch.from_numpy(_rng(seed, name).standard_normal(shape).astype(np.float32) * np.float32(std))


def _uniform(seed, name, shape, lo, hi):
    return torch.from_numpy(_rng(seed, name).uniform(lo, hi, shape).astype(np.float32))


def _bn(sd, seed, prefix, ch, gain):
    # FrozenBatchNorm2d buffers: layers/normalization.py:11-16
    sd[prefix + '.weight'] = _uniform(seed, prefix + '.weight', (ch,), 0.8, 1.2) * gain
    sd[prefix + '.bias'] = _normal(seed, prefix + '.bias', (ch,), 0.05)
    sd[prefix + '.running_mean'] = _normal(seed, prefix + '.running_mean', (ch,), 0.05)
    sd[prefix + '.running_var'] = _uniform(seed, prefix + '.running_var', (ch,), 0.8, 1.2)


def _conv(sd, seed, name, cout, cin, k, gain=1.0):
    std = gain * math.sqrt(2.0 / (cin * k * k))
    sd[name] = _normal(seed, name, (cout, cin, k, k), std)


def _linear(sd, seed, prefix, cout, cin, gain=1.0, bias_std=0.02):
    bound = gain * math.sqrt(6.0 / (cin + cout))  # xavier_uniform, transformer.py:694-697
    sd[prefix + '.weight'] = _uniform(seed, prefix + '.weight', (cout, cin), -bound, bound)
    sd[prefix + '.bias'] = _normal(seed, prefix + '.bias', (cout,), bias_std)


def _norm(sd, seed, prefix, ch):
    sd[prefix + '.weight'] = _uniform(seed, prefix + '.weight', (ch,), 0.9, 1.1)
    sd[prefix + '.bias'] = _normal(seed, prefix + '.bias', (ch,), 0.02)


SWIN_B = dict(embed_dim=128, depths=(2, 2, 18), heads=(4, 8, 16), window=7)   # encoders/swin/build.py:11-22, last stage dropped


def swin_relative_position_index(ws: int = 7) -> torch.Tensor:
    """encoders/swin/swin_transformer.py:128-144: index into the (2w-1)^2 bias table for every token pair of a window."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def _swin_encoder(sd, seed):
    e, depths, heads, ws = SWIN_B['embed_dim'], SWIN_B['depths'], SWIN_B['heads'], SWIN_B['window']
    _conv(sd, seed, 'encoder.patch_embed.proj.weight', e, 3, 4)
    sd['encoder.patch_embed.proj.bias'] = _normal(seed, 'encoder.patch_embed.proj.bias', (e,), 0.02)
    _norm(sd, seed, 'encoder.patch_embed.norm', e)
    for li, (depth, nh) in enumerate(zip(depths, heads)):
        dim = e * 2 ** li
        for b in range(depth):
            p = f'encoder.layers.{li}.blocks.{b}'
            _norm(sd, seed, p + '.norm1', dim)
            sd[p + '.attn.relative_position_bias_table'] = _normal(seed, p + '.attn.relative_position_bias_table', ((2 * ws - 1) ** 2, nh), 0.5)
            sd[p + '.attn.relative_position_index'] = swin_relative_position_index(ws)
            _linear(sd, seed, p + '.attn.qkv', 3 * dim, dim, 1.6)
            _linear(sd, seed, p + '.attn.proj', dim, dim, 0.7)
            _norm(sd, seed, p + '.norm2', dim)
            _linear(sd, seed, p + '.mlp.fc1', 4 * dim, dim)
            _linear(sd, seed, p + '.mlp.fc2', dim, 4 * dim, 0.7)
        if li < len(depths) - 1:
            p = f'encoder.layers.{li}.downsample'
            bound = math.sqrt(6.0 / (4 * dim + 2 * dim))
            sd[p + '.reduction.weight'] = _uniform(seed, p + '.reduction.weight', (2 * dim, 4 * dim), -bound, bound)
            _norm(sd, seed, p + '.norm', 4 * dim)
    for li in range(len(depths)):
        _norm(sd, seed, f'encoder.norm{li}', e * 2 ** li)


def _deaot_gpm(sd, seed, num_lstt, d_model):
    """DualBranchGPM (layers/transformer.py:700-763) of R50-DeAOTL: att/self heads 1, d_att = d_model / 2, expand ratio 2
    (transformer.py:1011-1082; layers/attention.py:93-136, 220-279)."""
    d_att, e1, e2 = d_model // 2, 2 * d_model, 4 * d_model
    for i in range(num_lstt):
        p = f'LSTT.layers.{i}'
        _norm(sd, seed, p + '.norm1', d_model)
        _linear(sd, seed, p + '.linear_QV', d_att + e1, d_model, 1.6)
        _linear(sd, seed, p + '.linear_U', e1, d_model, 1.6)
        if i == 0:
            _linear(sd, seed, p + '.linear_ID_V', e1, d_model, 1.6)
        else:
            _norm(sd, seed, p + '.id_norm1', d_model)
            _linear(sd, seed, p + '.linear_ID_V', e1, 2 * d_model, 1.6)
            _linear(sd, seed, p + '.linear_ID_U', e1, d_model, 1.6)
        for nm in ('long_term_attn', 'short_term_attn'):
            if nm == 'short_term_attn':
                sd[f'{p}.{nm}.relative_emb_k.weight'] = _normal(seed, f'{p}.{nm}.relative_emb_k.weight', (225, d_att, 1, 1), 0.08)
                sd[f'{p}.{nm}.relative_emb_k.bias'] = _normal(seed, f'{p}.{nm}.relative_emb_k.bias', (225,), 0.3)
            sd[f'{p}.{nm}.dw_conv.conv.weight'] = _normal(seed, f'{p}.{nm}.dw_conv.conv.weight', (e2, 1, 5, 5), 0.25)
            _linear(sd, seed, f'{p}.{nm}.projection', e1, e2, 1.6)
        _norm(sd, seed, p + '.norm2', d_model)
        _norm(sd, seed, p + '.id_norm2', d_model)
        _linear(sd, seed, p + '.self_attn.linear_QK', d_att, e1, 1.6)
        for nm in ('linear_V1', 'linear_V2', 'linear_U1', 'linear_U2'):
            _linear(sd, seed, f'{p}.self_attn.{nm}', e1, d_model, 1.6)
        sd[p + '.self_attn.dw_conv.conv.weight'] = _normal(seed, p + '.self_attn.dw_conv.conv.weight', (e2, 1, 5, 5), 0.25)
        _linear(sd, seed, p + '.self_attn.projection', e1, e2, 1.6)
    _norm(sd, seed, 'LSTT.decoder_norms.0.gn', 2 * d_model)      # final GroupNorm1D(512, 2), transformer.py:755-758


def synth_state_dict(seed: int = 0, num_lstt: int = 3, d_model: int = 256,
                     max_obj_num: int = 10, encoder: str = 'resnet50', model: str = 'aot') -> "OrderedDict[str, torch.Tensor]":
    """All tensors of the R50-AOTL (362), SwinB-AOTL (471) or R50-DeAOTL (356, model='deaot') state_dict, fp32, CPU."""
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    swin = encoder == 'swin_base'
    deaot = model == 'deaot'
    enc_dims = (128, 256, 512) if swin else (256, 512, 1024)
    # learned temporal positional embedding, aot.py:95-103 / deaot.py:46-53 (half width) (std raised from .05 so the
    # path is visible in parity tests)
    pe_dim = d_model // 2 if deaot else d_model
    sd['cur_pos_emb'] = _normal(seed, 'cur_pos_emb', (1, pe_dim), 0.3)
    sd['mem_pos_emb'] = _normal(seed, 'mem_pos_emb', (4, pe_dim), 0.3)

    # --- encoder (torchvision-style names) ---
    if swin:
        _swin_encoder(sd, seed)
    else:
        _conv(sd, seed, 'encoder.conv1.weight', 64, 3, 7)
        _bn(sd, seed, 'encoder.bn1', 64, 1.0)
    for idx, planes, blocks, inplanes in (() if swin else _R50_STAGES):
        for b in range(blocks):
            p = f'encoder.layer{idx}.{b}'
            cin = inplanes if b == 0 else planes * 4
            _conv(sd, seed, p + '.conv1.weight', planes, cin, 1)
            _bn(sd, seed, p + '.bn1', planes, 1.0)
            _conv(sd, seed, p + '.conv2.weight', planes, planes, 3)
            _bn(sd, seed, p + '.bn2', planes, 1.0)
            _conv(sd, seed, p + '.conv3.weight', planes * 4, planes, 1)
            _bn(sd, seed, p + '.bn3', planes * 4, 0.45)
            if b == 0:
                _conv(sd, seed, p + '.downsample.0.weight', planes * 4, cin, 1)
                _bn(sd, seed, p + '.downsample.1', planes * 4, 0.7)
    # encoder_projector 1x1 1024->256, aot.py:25-29
    sd['encoder_projector.weight'] = _normal(seed, 'encoder_projector.weight', (d_model, enc_dims[2], 1, 1),
                                             0.06 if swin else 0.042)
    sd['encoder_projector.bias'] = _normal(seed, 'encoder_projector.bias', (d_model,), 0.02)

    # --- LSTT ---
    if deaot:
        _deaot_gpm(sd, seed, num_lstt, d_model)
    for i in range(0 if deaot else num_lstt):
        p = f'LSTT.layers.{i}'
        _norm(sd, seed, p + '.norm1', d_model)
        for nm in ('linear_Q', 'linear_K', 'linear_V', 'projection'):
            _linear(sd, seed, f'{p}.self_attn.{nm}', d_model, d_model, 1.3 if nm in ('linear_Q', 'linear_K') else 1.0)
        _norm(sd, seed, p + '.norm2', d_model)
        _linear(sd, seed, p + '.linear_Q', d_model, d_model, 1.5)
        _linear(sd, seed, p + '.linear_V', d_model, d_model)
        _linear(sd, seed, p + '.linear_QMem', d_model, d_model)
        _linear(sd, seed, p + '.linear_VMem', d_model, d_model)
        _norm(sd, seed, p + '.norm4', d_model)
        _linear(sd, seed, p + '.linear_KMem', d_model, d_model)  # unused on the path (transformer.py:494)
        _linear(sd, seed, p + '.long_term_attn.projection', d_model, d_model)
        _linear(sd, seed, p + '.short_term_attn.projection', d_model, d_model)
        _norm(sd, seed, p + '.norm3', d_model)
        _linear(sd, seed, p + '.linear1', 4 * d_model, d_model)
        _norm(sd, seed, p + '.activation.gn', 4 * d_model)
        sd[p + '.activation.conv.weight'] = _normal(seed, p + '.activation.conv.weight', (4 * d_model, 1, 5, 5), 0.2)
        _linear(sd, seed, p + '.linear2', d_model, 4 * d_model)
    for i in range(0 if deaot else num_lstt):
        _norm(sd, seed, f'LSTT.decoder_norms.{i}', d_model)

    # --- FPN decoder, decoders/fpn.py:22-32 ---
    def convgn(prefix, cout, cin, k):
        _conv(sd, seed, prefix + '.conv.weight', cout, cin, k, 0.7)
        sd[prefix + '.conv.bias'] = _normal(seed, prefix + '.conv.bias', (cout,), 0.02)
        _norm(sd, seed, prefix + '.gn', cout)
    convgn('decoder.conv_in', d_model, 2 * d_model if deaot else d_model * (num_lstt + 1), 1)   # deaot.py:28-31
    convgn('decoder.conv_16x', d_model, d_model, 3)
    convgn('decoder.conv_8x', d_model // 2, d_model, 3)
    convgn('decoder.conv_4x', d_model // 2, d_model // 2, 3)
    for nm, cout, cin in (('adapter_16x', d_model, enc_dims[2]), ('adapter_8x', d_model, enc_dims[1]), ('adapter_4x', d_model // 2, enc_dims[0])):
        _conv(sd, seed, f'decoder.{nm}.weight', cout, cin, 1, 0.7)
        sd[f'decoder.{nm}.bias'] = _normal(seed, f'decoder.{nm}.bias', (cout,), 0.02)
    _conv(sd, seed, 'decoder.conv_out.weight', max_obj_num + 1, d_model // 2, 1, 1.0)
    sd['decoder.conv_out.bias'] = _normal(seed, 'decoder.conv_out.bias', (max_obj_num + 1,), 0.02)

    # --- identity bank Conv2d(12->256, k17, s16, p8), aot.py:68-74 ---
    kid = 16 if swin else 17      # models/aot.py:67-82: k17 s16 p8 (align_corners) or k16 s16 p0
    sd['patch_wise_id_bank.weight'] = _normal(seed, 'patch_wise_id_bank.weight', (d_model, max_obj_num + 2, kid, kid), 1.0 / kid)
    sd['patch_wise_id_bank.bias'] = _normal(seed, 'patch_wise_id_bank.bias', (d_model,), 0.02)
    if deaot:
        _norm(sd, seed, 'id_norm', d_model)          # deaot.py:42
    return sd


TRAINED_DELTA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'trained_delta.pt')


def fitted_state_dict(seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """synth_state_dict(seed) with every non-encoder tensor replaced by the values fitted on synthetic clips
    (tests/golden/train_synth_weights.py -> tests/golden/trained_delta.pt, bf16).  The random ResNet-50 stays; the LSTT,
    decoder, identity bank and temporal PE were trained to propagate masks, so logits are confident ("trained-like")."""
    sd = synth_state_dict(seed)
    delta = torch.load(TRAINED_DELTA, map_location='cpu')
    for k, v in delta.items():
        assert k in sd and sd[k].shape == v.shape, k
        sd[k] = v.float()
    return sd
