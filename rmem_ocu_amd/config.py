"""Configuration object with the reference's UPPER_CASE attribute names.

Only the attributes the inference path reads are kept:
configs/models/default.py:3-26 overlaid by configs/models/r50_aotl.py:7-39 (or r50_deaotl.py / swinb_aotl.py) and the
``pre_vost`` stage (configs/pre_vost.py:14-18), plus the overrides tools/eval.py:108-135
applies from the command line.
"""
from __future__ import annotations


class EngineConfig:
    def __init__(self, exp_name: str = 'default', model: str = 'r50_aotl'):
        model = model.lower()
        if model not in ('r50_aotl', 'swinb_aotl', 'r50_deaotl'):
            raise NotImplementedError(f'model config {model!r}: built configs are r50_aotl, swinb_aotl and r50_deaotl')
        self.EXP_NAME = exp_name
        self.MODEL_NAME = 'R50_AOTL_Temp_pe_Slot_4'
        self.MODEL_VOS = 'aot'
        self.MODEL_ENGINE = 'aotengine'
        self.MODEL_ALIGN_CORNERS = True
        self.MODEL_ENCODER = 'resnet50'
        self.MODEL_ENCODER_DIM = [256, 512, 1024, 1024]
        self.MODEL_ENCODER_EMBEDDING_DIM = 256
        self.MODEL_DECODER_INTERMEDIATE_LSTT = True
        self.MODEL_LINEAR_Q = False          # stage pre_vost (configs/pre_vost.py:16)
        self.MODEL_FREEZE_BN = True
        self.MODEL_MAX_OBJ_NUM = 10
        self.MODEL_IGNORE_TOKEN = True
        self.MODEL_SELF_HEADS = 8
        self.MODEL_ATT_HEADS = 8
        self.MODEL_LSTT_NUM = 3
        self.MODEL_EPSILON = 1e-5
        self.MODEL_DTYPE = 'bf16'            # build-specific: 16-bit operand type of the HIP kernels, 'bf16' | 'fp16' (the reference: --amp)
        self.FORMER_MEM_LEN = 1
        self.LATTER_MEM_LEN = 7              # "N = 8" <=> 1 + 7
        self.USE_TEMPORAL_POSITIONAL_EMBEDDING = True
        self.TEMPORAL_POSITIONAL_EMBEDDING_SLOT_4 = True
        self.GRU_MEMORY = False
        self.TIME_ENCODE = False
        self.TIME_ENCODE_NORM = False
        self.USE_MASK = False
        self.NO_LONG_MEMORY = False
        self.NO_MEMORY_GAP = False
        self.REVERSE_INFER = False
        self.TEST_LONG_TERM_MEM_GAP = 5
        self.TEST_MAX_SIZE = 800 * 1.3
        self.TEST_FLIP = False
        self.TEST_MULTISCALE = [1.0]
        if model == 'swinb_aotl':
            # configs/models/swinb_aotl.py:8-13 overlaid on the RMem attributes above (the reference's own swinb config lacks
            # them, so it cannot build an AOT model as shipped: SURVEY.md §8 a16)
            self.MODEL_NAME = 'SwinB_AOTL_Temp_pe_Slot_4'
            self.MODEL_ENCODER = 'swin_base'
            self.MODEL_ALIGN_CORNERS = False
            self.MODEL_ENCODER_DIM = [128, 256, 512, 512]
        if model == 'r50_deaotl':
            # configs/models/default_deaot.py:7-18 + configs/models/r50_deaotl.py:8-40
            self.MODEL_NAME = 'R50_DeAOTL_Temp_pe_Slot_4'
            self.MODEL_VOS = 'deaot'
            self.MODEL_ENGINE = 'deaotengine'
            self.MODEL_DECODER_INTERMEDIATE_LSTT = False
            self.MODEL_SELF_HEADS = 1
            self.MODEL_ATT_HEADS = 1
            self.LATTER_MEM_LEN = 8


def get_config(stage: str = 'pre_vost', exp_name: str = 'default', model: str = 'r50_aotl') -> EngineConfig:
    """tools/get_config.py:4-6."""
    return EngineConfig(exp_name, model)
