"""ctypes binding of librmem_hip.so (include/rmem.h).

There is no CPU fallback: if the shared object is missing or a call fails, an
exception is raised.  ``build()`` compiles it in-tree with hipcc for gfx950.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RMEM_LIB_PATH') or os.path.join(_HERE, 'librmem_hip.so')   # override: kernel experiments only
ABI_VERSION = 8


class RmemError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('H', 'W', 'Cin', 'Ho', 'Wo', 'Cout', 'KH', 'KW', 'stride', 'pad',
                                       'ldo', 'ldr', 'ld2', 'relu', 'out_f32', 'res_f32', 'ldx', 'batch', 'act_begin',
                                       'res_up_h', 'res_up_w', 'res_up_align')]


class BneckChainDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('batch', 'Ho', 'Wo', 'K1', 'Cout', 'N2', 'H2', 'W2', 'Cin2', 'stride2')]


class AttnChunk(C.Structure):
    _fields_ = [('slot', C.c_int), ('key_begin', C.c_int), ('key_count', C.c_int), ('pe_slot', C.c_int),
                ('t', C.c_int), ('reserved', C.c_int * 3)]


_vp, _i, _ll, _f = C.c_void_p, C.c_int, C.c_longlong, C.c_float


def _desc(name, ints, ptrs):
    """ctypes mirror of an rmem_chain_*_desc: (int, int, float, int) header + device pointers, in include/rmem.h's order."""
    return type(name, (C.Structure,), {'_fields_': [(ints[0], _i), (ints[1], _i), ('eps', _f), (ints[2], _i)] + [(n, _vp) for n in ptrs]})


ChainA = _desc('ChainA', ('L', 'clips', 'reserved'),
               ('att', 'x', 'w_proj', 'b_proj', 'ln2_g', 'ln2_b', 'curr_v', 'w_q', 'b_q', 'curr_q', 'short_k', 'short_v', 'ln4_g', 'ln4_b',
                'k4', 'v4'))
ChainB = _desc('ChainB', ('L', 'clips', 'gn_splits'),
               ('att_long', 'att_short', 'x', 'w_long', 'b_long', 'w_short', 'b_short', 'tgt3', 'ln3_g', 'ln3_b', 'w1', 'b1', 'h1',
                'gn_partial'))
ChainC = _desc('ChainC', ('L', 'clips', 'ld_dec'),
               ('x', 'h3', 'w2', 'b2', 'dec_g', 'dec_b', 'dec_out', 'ln1_g', 'ln1_b', 'w_qkv', 'b_qkv', 'pos_qk', 'qkv'))
# name -> (restype, argtypes); the list is checked against include/rmem.h by tests/test_abi.py
SIGNATURES = {
    'rmem_abi_version': (_i, []),
    'rmem_last_error_string': (C.c_char_p, []),
    'rmem_conv_workspace_bytes': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'rmem_conv2d_nhwc': (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'rmem_attn_workspace_bytes': (C.c_size_t, [_i, _i, _i]),
    'rmem_mem_read_attn': (_i, [_vp, _i, _vp, _vp, _ll, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    'rmem_profile_start': (_i, [_i]),
    'rmem_profile_stop': (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    'rmem_layernorm256': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _f, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp]),
    'rmem_layernorm': (_i, [_vp, _i, _i, _vp, _vp, _f, _i, _i, _vp, _i, _vp, _i, _vp]),
    'rmem_patch_merge_ln': (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _vp, _vp]),
    'rmem_window_attn': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'rmem_window_attn_images': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'rmem_patch_merge_ln_images': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _vp, _vp]),
    'rmem_add16': (_i, [_vp, _vp, _vp, _ll, _vp]),
    'rmem_add16_grouped': (_i, [_i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _ll, _vp]),
    'rmem_layernorm256_pair': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp]),
    'rmem_lstt_chain_waves': (_i, []),
    'rmem_lstt_chain_a': (_i, [C.POINTER(ChainA), _vp]),
    'rmem_lstt_chain_b': (_i, [C.POINTER(ChainB), _vp]),
    'rmem_lstt_chain_c': (_i, [C.POINTER(ChainC), _vp]),
    'rmem_conv1x1_dual_nhwc': (_i, [C.POINTER(ConvDesc), _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rmem_bneck_chain': (_i, [C.POINTER(BneckChainDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'rmem_linear_grouped': (_i, [C.POINTER(ConvDesc), _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    'rmem_groupnorm_workspace_bytes': (C.c_size_t, [_i]),
    'rmem_groupnorm_nhwc': (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    'rmem_groupnorm_f32_nhwc': (_i, [_vp, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    'rmem_groupnorm_nhwc_images': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp]),
    'rmem_groupnorm_head_nhwc_images': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _i, _vp, _i, _vp, _vp]),
    'rmem_gn_act_dwconv5x5_nhwc_images': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    'rmem_gn_act_dwconv5x5_prestats_nhwc_images': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    'rmem_bilinear_nhwc_images': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'rmem_mem_read_attn_clips': (_i, [_vp, _i, _vp, _vp, _ll, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _ll, _ll, _ll, _vp, _vp]),
    'rmem_lstt_attn_pair_clips': (_i, [_vp, _i, _vp, _vp, _ll, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _ll, _ll,
                                       _vp, _vp, _i, _ll, _vp, _ll, _vp, _vp]),
    'rmem_gn_act_dwconv5x5_nhwc': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp]),
    'rmem_dwconv5x5_nhwc': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'rmem_image_to_nhwc8': (_i, [_vp, _vp, _i, _i, _vp]),
    'rmem_image_to_nhwc8_images': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'rmem_image_ptrs_to_nhwc8': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'rmem_stem_padded_size': (_i, [_i, _i, C.POINTER(_i), C.POINTER(_i)]),
    'rmem_image_ptrs_to_nhwc4p': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'rmem_stem7x7s2': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rmem_stem7x7s2_pool': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rmem_conv3x3_c64_direct': (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rmem_conv3x3_direct': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    'rmem_ingest_rgb8': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    'rmem_maxpool3x3s2_nhwc': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'rmem_maxpool3x3s2_nhwc_images': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'rmem_bilinear_nhwc': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'rmem_logits_post': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rmem_logits_post_images': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rmem_label_to_onehot16': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'rmem_label_id_embed_scratch_size': (_i, [_i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    'rmem_label_id_embed': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'rmem_label_to_onehot16_images': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'rmem_evict_scores': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp]),
    'rmem_resize_nearest_flip_f32': (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _vp]),
    'rmem_tta_merge': (_i, [C.POINTER(_vp), C.POINTER(_i), _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'rmem_mask_iou_counts': (_i, [_vp, _vp, _ll, _i, _i, _vp, _vp]),
    'rmem_split_label': (_i, [_vp, _i, _i, _vp, _ll, _vp]),
    'rmem_soft_logit_aggregate': (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp]),
    'rmem_copy_async': (_i, [_vp, _vp, C.c_size_t, _vp]),
    'rmem_scatter_blocks': (_i, [_vp, _vp, _vp, _i, _ll, _ll, _vp]),
    'rmem_copy2d_async': (_i, [_vp, _ll, _vp, _ll, _ll, _i, _vp]),
    'rmem_gated_attn_workspace_bytes': (C.c_size_t, [_i, _i, _i, _i, _i]),
    'rmem_gated_attn': (_i, [_vp, _i, _vp, _ll, _i, _vp, _ll, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i,
                             _vp, _vp, _i, _i, _vp, _vp]),
    'rmem_local_gated_attn': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    'rmem_gated_attn_clips': (_i, [_vp, _i, _vp, _ll, _i, _vp, _ll, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i,
                                   _vp, _vp, _i, _i, _i, _vp, _vp]),
    'rmem_local_gated_attn_clips': (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    'rmem_gated_profile_start': (_i, []),
    'rmem_gated_profile_stop': (_i, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    'rmem_graph_begin': (_i, [_vp]),
    'rmem_graph_end': (_i, [_vp, C.POINTER(_vp)]),
    'rmem_graph_launch': (_i, [_vp, _vp]),
    'rmem_graph_destroy': (_i, [_vp]),
}

# entry points with 16-bit operands exist twice: <name> (bfloat16) and <name>_f16 (IEEE half), same signature (include/rmem.h)
F16_TWINS = ('rmem_conv2d_nhwc', 'rmem_mem_read_attn', 'rmem_mem_read_attn_clips', 'rmem_lstt_attn_pair_clips', 'rmem_layernorm256', 'rmem_layernorm', 'rmem_patch_merge_ln',
             'rmem_window_attn', 'rmem_window_attn_images', 'rmem_patch_merge_ln_images', 'rmem_add16', 'rmem_add16_grouped', 'rmem_layernorm256_pair', 'rmem_lstt_chain_a', 'rmem_lstt_chain_b', 'rmem_lstt_chain_c', 'rmem_conv1x1_dual_nhwc', 'rmem_bneck_chain', 'rmem_label_id_embed', 'rmem_image_ptrs_to_nhwc4p', 'rmem_stem7x7s2', 'rmem_stem7x7s2_pool', 'rmem_conv3x3_c64_direct', 'rmem_conv3x3_direct', 'rmem_linear_grouped',
             'rmem_groupnorm_nhwc', 'rmem_groupnorm_f32_nhwc', 'rmem_groupnorm_nhwc_images', 'rmem_groupnorm_head_nhwc_images',
             'rmem_gn_act_dwconv5x5_nhwc_images', 'rmem_gn_act_dwconv5x5_prestats_nhwc_images', 'rmem_gn_act_dwconv5x5_nhwc', 'rmem_dwconv5x5_nhwc', 'rmem_image_to_nhwc8',
             'rmem_image_to_nhwc8_images', 'rmem_image_ptrs_to_nhwc8', 'rmem_ingest_rgb8', 'rmem_maxpool3x3s2_nhwc', 'rmem_maxpool3x3s2_nhwc_images', 'rmem_bilinear_nhwc',
             'rmem_bilinear_nhwc_images', 'rmem_label_to_onehot16', 'rmem_label_to_onehot16_images', 'rmem_gated_attn', 'rmem_local_gated_attn', 'rmem_gated_attn_clips', 'rmem_local_gated_attn_clips')
SIGNATURES.update({n + '_f16': SIGNATURES[n] for n in F16_TWINS})

_lib = None


def build(force: bool = False) -> str:
    """Compile librmem_hip.so in-tree (hipcc --offload-arch=gfx950)."""
    src = os.path.join(_HERE, 'csrc')
    if force:
        subprocess.run(['make', '-C', src, 'clean'], check=True, capture_output=True)
    r = subprocess.run(['make', '-C', src, '-j8'], capture_output=True, text=True)
    if r.returncode != 0:
        raise RmemError('building librmem_hip.so failed:\n' + r.stdout + r.stderr)
    return LIB_PATH


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RmemError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                            '(there is no CPU fallback for the HIP path)')
        # torch bundles its own libamdhip64; it must be in the process first so that this library binds to the
        # SAME HIP runtime (streams and device pointers are torch's), not to a second copy from /opt/rocm
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.rmem_abi_version() != ABI_VERSION:
            raise RmemError('librmem_hip.so ABI version mismatch; rebuild it')
        _lib = L
    return _lib


def check(rc: int, what: str = ''):
    if rc != 0:
        raise RmemError(f'{what} failed ({rc}): {lib().rmem_last_error_string().decode()}')
