"""Pins oracle/deaot_cpu.py (R50-DeAOTL path) against fixtures produced by the reference itself
(tests/golden/make_golden.py deaot).  CPU only; the reference is NOT needed at run time."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import deaot_cpu as D
from oracle import ref_cpu as O

from conftest import GOLDEN
from test_oracle_golden import _check_clip, close, seeded


@pytest.fixture(scope='module')
def w():
    from rmem_ocu_amd.weights import synth_state_dict
    return synth_state_dict(0, model='deaot')


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(GOLDEN, 'deaot_ops.npz'))


def test_state_dict_contract(w):
    assert len(w) == 356
    assert w['cur_pos_emb'].shape == (1, 128) and w['mem_pos_emb'].shape == (4, 128)
    assert w['LSTT.layers.0.linear_ID_V.weight'].shape == (512, 256)
    assert w['LSTT.layers.1.linear_ID_V.weight'].shape == (512, 512)
    assert w['decoder.conv_in.conv.weight'].shape == (256, 512, 1, 1)


@pytest.mark.parametrize('T', [1, 4, 9])
def test_gated_propagation(g, w, T):
    h, wd = 9, 11
    L = h * wd
    q, k = seeded(4000 + T, (L, 1, 128)), seeded(4100 + T, (T * L, 1, 128))
    v, u = seeded(4200 + T, (T * L, 1, 1024)), seeded(4300 + T, (L, 1, 1024))
    o, attn = D.gated_propagation(q, k, v, u, (h, wd), w, 'LSTT.layers.0.long_term_attn', False, True)
    close(o[:, 0], g[f'gp_T{T}_out'])
    close(attn.view(1, 1, L, T, L).mean(1)[0].sum(2), g[f'gp_T{T}_mass'])
    o2, _ = D.gated_propagation(q, k, v, u, (h, wd), w, 'LSTT.layers.0.long_term_attn', False, False)   # SDPA flavour
    close(o2[:, 0], g[f'gp_T{T}_out'])


def test_local_gated_propagation(g, w):
    h, wd = 9, 11
    o = D.local_gated_propagation(seeded(4400, (1, 128, h, wd)), seeded(4401, (1, 128, h, wd)), seeded(4402, (1, 1024, h, wd)),
                                  seeded(4403, (h * wd, 1, 1024)), (h, wd), w, 'LSTT.layers.0.short_term_attn')
    close(o[:, 0], g['lgp_out'])
    h, wd = 18, 23
    o = D.local_gated_propagation(seeded(4410, (1, 128, h, wd)), seeded(4411, (1, 128, h, wd)), seeded(4412, (1, 1024, h, wd)),
                                  seeded(4413, (h * wd, 1, 1024)), (h, wd), w, 'LSTT.layers.0.short_term_attn')
    close(o[::2, 0], g['lgp_big_out'])


def test_self_gated_propagation(g, w):
    x = seeded(4500, (99, 1, 512))
    o, _ = D.gated_propagation(x, x, x, x, (9, 11), w, 'LSTT.layers.0.self_attn', True, False)
    close(o[:, 0], g['gp_self_out'])


@pytest.mark.parametrize('li', [0, 1])
@pytest.mark.parametrize('T', [1, 3, 5, 9])
def test_gpm_block(g, w, li, T):
    h, wd = 9, 11
    L, C = h * wd, 256
    temporal = torch.cat((w['cur_pos_emb'], w['mem_pos_emb']), 0)
    tgt = seeded(3000 + 10 * T + li, (L, 1, C))
    tgt_id = None if li == 0 else seeded(3100 + T, (L, 1, C))
    long_mem = [seeded(3200 + T, (T, L, 1, 128)), F.silu(seeded(3300 + T, (T, L, 1, 512))), None, F.silu(seeded(3400 + T, (T, L, 1, 512)))]
    short_mem = [seeded(3500 + T, (1, 128, h, wd)), F.silu(seeded(3600 + T, (1, 512, h, wd))), None, F.silu(seeded(3700 + T, (1, 512, h, wd)))]
    y, yid, mems, rec = D.gpm_block(tgt, tgt_id, w, f'LSTT.layers.{li}', long_mem, short_mem, None, (h, wd), temporal, True)
    close(y[::3, 0], g[f'gpm{li}_T{T}_out'])
    close(yid[::3, 0], g[f'gpm{li}_T{T}_outid'])
    close(mems[0][0][::3, 0], g[f'gpm{li}_T{T}_curK'])
    close(mems[0][1][::3, 0], g[f'gpm{li}_T{T}_curV'])
    close(rec, g[f'gpm{li}_T{T}_mass'])


@pytest.mark.parametrize('li', [0, 1])
def test_gpm_block_reference_frame(g, w, li):
    h, wd = 9, 11
    temporal = torch.cat((w['cur_pos_emb'], w['mem_pos_emb']), 0)
    tgt = seeded(3800 + li, (99, 1, 256))
    tgt_id = None if li == 0 else seeded(3810, (99, 1, 256))
    y, yid, mems, _ = D.gpm_block(tgt, tgt_id, w, f'LSTT.layers.{li}', None, None, seeded(3820, (99, 1, 256), 0.5), (h, wd), temporal, False)
    close(y[:, 0], g[f'gpm{li}_ref_out'])
    close(yid[:, 0], g[f'gpm{li}_ref_outid'])
    close(mems[1][3][0, :, 0], g[f'gpm{li}_ref_gIDV'])


def test_id_emb_and_decoder(g, w):
    xs = O.encode_image(seeded(1100, (1, 3, 97, 129)), w)
    embs = [seeded(4600 + i, (63, 1, 512)) for i in range(3)]
    close(O.fpn_decode([embs[-1].view(7, 9, 1, 512).permute(2, 3, 0, 1)], xs, w)[0], g['dec_logits'])
    mask = torch.zeros(1, 1, 97, 129, dtype=torch.int32)
    mask[:, :, 10:50, 20:70] = 1
    mask[:, :, 40:90, 60:120] = 3
    oh, _ = O.one_hot_mask(mask)
    eng = D.OracleDeAOTEngine(w)
    close(eng._assign_identity(oh, None).permute(1, 2, 0).reshape(256, 7, 9), g['id_emb'])


def test_small_clip_matches_reference():
    """48 frames, bank 1 + 2, gap 2: masks, eviction trace (scores move on every long-term update) and logits."""
    _check_clip('deaot_clip_small.npz')


def test_full_clip_matches_reference():
    """cfg-2 geometry (481x849 -> 31x54 tokens, larger than the 15x15 window), bank 1 + 8, gap 2."""
    _check_clip('deaot_clip_full.npz')


def test_new_object_raises_like_the_reference():
    """R50-DeAOTL with a mid-clip reference frame (a new object's mask, managers/evaluator.py:484-508): the reference raises at the
    first long-term update afterwards -- DualBranchGPM.restrict_long_memories has no early return while the bank is not full
    (layers/transformer.py:880-892) and long_memories_indexes kept growing across the bank reset (aot_engine.py:322-323); observed
    by running the reference (tests/golden/make_golden.py gen_clip(..., inject_at=5, model_name='r50_deaotl') -> RuntimeError).  The
    restatement reproduces the failure instead of inventing a behaviour."""
    import pytest
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    frames, mask = make_clip(31, 10, 97, 129, 2)
    eng = D.OracleDeAOTEngine(synth_state_dict(0, model='deaot'), 1, 8, 2)
    eng.add_reference_frame(frames[0:1], mask, 0)
    with pytest.raises(RuntimeError):
        for i in range(1, 10):
            logit = eng.match_propogate_one_frame(frames[i:i + 1], (96, 128))
            label = torch.argmax(logit, 1, keepdim=True).float()
            m = torch.nn.functional.interpolate(label, size=eng.input_size_2d, mode='nearest')
            if i == 4:
                eng.add_reference_frame(frames[i:i + 1], m, i)
            else:
                eng.update_memory(m)
