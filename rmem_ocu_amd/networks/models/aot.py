"""AOT model object: the reference's parameter tree, executed by the HIP runtime.

Mirrors networks/models/aot.py:12-105 at the interface level: the same attributes
(``encoder, encoder_projector, LSTT, decoder, patch_wise_id_bank, cur_pos_emb,
mem_pos_emb, cfg, max_obj_num``) and the same 362 ``state_dict`` keys, so a reference
checkpoint loads with ``load_state_dict`` / utils/checkpoint.py:75-104 unchanged.
The modules are parameter containers only: no torch op runs in ``forward``; the
per-frame math is the launch lists of rmem_ocu_amd.runtime.ClipRuntime over
``packed()`` (BN-folded, bf16, NHWC weights).  Unlike the reference (which keeps the
clip's LSTT memory inside the shared model, layers/transformer.py:455-463), clip
state lives in the engine, so one weight set serves many clips per GPU.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from ...pack import pack_state_dict
from ...weights import synth_state_dict

_BUFFER_SUFFIXES = ('running_mean', 'running_var')


class _Node(nn.Module):
    """Nested parameter container addressed by the reference's dotted key names."""

    def __getitem__(self, idx):
        return getattr(self, str(idx))


def _attach(root: nn.Module, key: str, value: torch.Tensor):
    parts = key.split('.')
    node = root
    for p in parts[:-1]:
        if not hasattr(node, p):
            node.add_module(p, _Node())
        node = getattr(node, p)
    leaf = parts[-1]
    # FrozenBatchNorm2d keeps all four tensors as buffers (layers/normalization.py:13-16)
    is_bn = any(k in key for k in ('.bn1.', '.bn2.', '.bn3.', '.downsample.1.')) and key.startswith('encoder.')
    is_bn = is_bn or leaf == 'relative_position_index'       # Swin: an int64 buffer (swin_transformer.py:143)
    if leaf in _BUFFER_SUFFIXES or is_bn:
        node.register_buffer(leaf, value.clone())
    else:
        node.register_parameter(leaf, nn.Parameter(value.clone(), requires_grad=False))


class AOT(nn.Module):
    KIND = 'aot'

    def __init__(self, cfg, encoder='resnet50', decoder='fpn'):
        super().__init__()
        if encoder not in ('resnet50', 'swin_base') or decoder != 'fpn':
            raise NotImplementedError('built encoders: resnet50 (R50-AOTL) and swin_base (SwinB-AOTL); decoder: fpn')
        if self.KIND == 'aot' and cfg.MODEL_LINEAR_Q:
            raise NotImplementedError('MODEL_LINEAR_Q=True: the reference eval path itself crashes there '
                                      '(layers/transformer.py:650-665); use the pre_vost setting False')
        self.cfg = cfg
        self.max_obj_num = cfg.MODEL_MAX_OBJ_NUM
        self.epsilon = cfg.MODEL_EPSILON
        self.use_temporal_pe = cfg.USE_TEMPORAL_POSITIONAL_EMBEDDING
        # same construction-time randomness contract as the reference: fresh weights unless loaded
        for k, v in synth_state_dict(0, cfg.MODEL_LSTT_NUM, cfg.MODEL_ENCODER_EMBEDDING_DIM, cfg.MODEL_MAX_OBJ_NUM, encoder,
                                     model=self.KIND).items():
            _attach(self, k, v)
        self._packed: Optional[Dict[str, torch.Tensor]] = None
        self._packed_device = None
        self._runtimes = {}

    # -- weights ------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._packed = None
        self._runtimes = {}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def element_dtype(self):
        """cfg.MODEL_DTYPE: 'bf16' (default) or 'fp16' -- the 16-bit operand type of every kernel (the reference's counterpart
        is running with or without --amp, tools/eval.py:45-47; fp16 is what its autocast uses)."""
        name = str(getattr(self.cfg, 'MODEL_DTYPE', 'bf16')).lower()
        if name in ('bf16', 'bfloat16'):
            return torch.bfloat16
        if name in ('fp16', 'f16', 'float16', 'half'):
            return torch.float16
        raise ValueError(f"cfg.MODEL_DTYPE = {name!r}: expected 'bf16' or 'fp16'")

    def packed(self) -> Dict[str, torch.Tensor]:
        dev = self.cur_pos_emb.device
        dt = self.element_dtype()
        if self._packed is None or self._packed_device != (dev, dt):
            if dev.type != 'cuda':
                raise RuntimeError('the HIP engine needs the model on a GPU: call model.cuda(gpu_id) '
                                   '(there is no CPU execution path in this package)')
            self._packed = pack_state_dict(self.state_dict(), dev, self.cfg.MODEL_LSTT_NUM, dtype=dt)
            self._packed_device = (dev, dt)
            self._runtimes = {}
        return self._packed

    def forward(self, *a, **k):
        raise RuntimeError('AOT is driven through build_engine(...): add_reference_frame / match_propogate_one_frame / update_memory')
