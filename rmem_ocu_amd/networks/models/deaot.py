"""DeAOT model object: the reference's parameter tree (networks/models/deaot.py:11-68), executed by the HIP runtime.

Same construction as .aot.AOT with the DualBranchGPM stack in place of the LSTT: 356 ``state_dict`` keys
(``LSTT.layers.i.{linear_QV, linear_U, linear_ID_V, linear_ID_U, long_term_attn, short_term_attn, self_attn, ...}``,
``LSTT.decoder_norms.0.gn``, ``id_norm``, 128-wide ``cur_pos_emb`` / ``mem_pos_emb``), so a reference R50-DeAOTL checkpoint
(eval_vost.sh:26) loads with ``load_state_dict`` unchanged.  Per-frame math: rmem_ocu_amd.runtime_deaot.DeAOTRuntime.
"""
from __future__ import annotations

from .aot import AOT


class DeAOT(AOT):
    KIND = 'deaot'

    def __init__(self, cfg, encoder='resnet50', decoder='fpn'):
        if encoder != 'resnet50':
            raise NotImplementedError('DeAOT is built with the resnet50 encoder (configs/models/r50_deaotl.py:29)')
        if cfg.MODEL_ATT_HEADS != 1 or cfg.MODEL_SELF_HEADS != 1 or cfg.MODEL_DECODER_INTERMEDIATE_LSTT:
            raise NotImplementedError('DeAOT path: one attention head, decoder on the last layer only '
                                      '(configs/models/default_deaot.py:13-16)')
        super().__init__(cfg, encoder, decoder)
