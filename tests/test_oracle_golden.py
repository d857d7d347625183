"""Pins oracle/ref_cpu.py against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only; the reference is NOT needed at run time."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from rmem_ocu_amd.synth import make_clip

from conftest import GOLDEN

TOL = 2e-5  # fp32 re-association only (same ATen kernels as the reference run)


def seeded(seed, shape, scale=1.0):
    rng = np.random.Generator(np.random.PCG64([seed, 0xC0FFEE]))
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * np.float32(scale))


def close(a, b, tol=TOL):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_sine_pos(golden_ops):
    close(O.sine_pos_emb(6, 7)[:, 0], golden_ops['sine_pos_6x7'], 1e-6)
    close(O.sine_pos_emb(31, 54)[[0, 53, 54, 800, 1673], 0], golden_ops['sine_pos_31x54_rows'], 1e-6)


def test_temporal_slots(golden_ops):
    for T in range(2, 33):
        assert O.temporal_slots(T) == golden_ops[f'slots_T{T}'].tolist(), T
    assert O.temporal_slots(1) == [0]


@pytest.mark.parametrize('T', [1, 2, 4, 5, 8, 9, 12])
def test_mha_and_block(golden_ops, synth_weights, T):
    w = synth_weights
    L, C = 42, 256
    q, k, v = seeded(100 + T, (L, 1, C)), seeded(200 + T, (T * L, 1, C)), seeded(300 + T, (T * L, 1, C))
    o, attn = O.mha(q, k, v, w, 'LSTT.layers.0.long_term_attn', False, True)
    close(o[:, 0], golden_ops[f'mha_T{T}_out'])
    close(attn.view(1, 8, L, T, L).mean(1)[0].sum(2), golden_ops[f'mha_T{T}_mass'])
    pos = O.sine_pos_emb(6, 7)
    temporal = torch.cat((w['cur_pos_emb'], w['mem_pos_emb']), 0)
    tgt = seeded(400 + T, (L, 1, C))
    long_mem = [seeded(500 + T, (T, L, 1, C)), seeded(600 + T, (T, L, 1, C))]
    short_mem = [seeded(700 + T, (L, 1, C)), seeded(800 + T, (L, 1, C))]
    y, mems, rec = O.lstt_block(tgt, w, 'LSTT.layers.0', long_mem, short_mem, None, pos, (6, 7), temporal, True)
    close(y[:, 0], golden_ops[f'blk_T{T}_out'])
    close(mems[0][0][:, 0], golden_ops[f'blk_T{T}_curK'])
    close(mems[2][0][:, 0], golden_ops[f'blk_T{T}_locK'])
    close(mems[2][1][:, 0], golden_ops[f'blk_T{T}_locV'])
    close(rec, golden_ops[f'blk_T{T}_mass'])


def test_block_reference_frame(golden_ops, synth_weights):
    w = synth_weights
    pos = O.sine_pos_emb(6, 7)
    temporal = torch.cat((w['cur_pos_emb'], w['mem_pos_emb']), 0)
    y, mems, _ = O.lstt_block(seeded(900, (42, 1, 256)), w, 'LSTT.layers.0', None, None,
                              seeded(901, (42, 1, 256), 0.5), pos, (6, 7), temporal, False)
    close(y[:, 0], golden_ops['blk_ref_out'])
    close(mems[1][1][0, :, 0], golden_ops['blk_ref_gV'])
    close(mems[2][1][:, 0], golden_ops['blk_ref_locV'])


def test_mha_cfg2_size(golden_ops, synth_weights):
    L, T, C = 1674, 8, 256
    q, k, v = seeded(1000, (L, 1, C)), seeded(1001, (T * L, 1, C)), seeded(1002, (T * L, 1, C))
    o, attn = O.mha(q, k, v, synth_weights, 'LSTT.layers.0.long_term_attn', False, True)
    close(o[golden_ops['mha_big_rows'], 0], golden_ops['mha_big_out'])
    close(attn.view(1, 8, L, T, L).mean(1)[0].sum(2), golden_ops['mha_big_mass'])


def test_encoder_idbank_decoder(golden_ops, synth_weights):
    w = synth_weights
    xs = O.encode_image(seeded(1100, (1, 3, 97, 129)), w)
    for i, x in enumerate(xs):
        close(x[0, :, ::3, ::3] if i < 2 else x[0], golden_ops[f'enc_x{i}'])
    embs = [seeded(1200 + i, (63, 1, 256)) for i in range(3)]
    ins = [xs[-1]] + [e.view(7, 9, 1, 256).permute(2, 3, 0, 1) for e in embs]
    close(O.fpn_decode(ins, xs, w)[0], golden_ops['dec_logits'])
    mask = torch.zeros(1, 1, 97, 129, dtype=torch.int32)
    mask[:, :, 10:50, 20:70] = 1
    mask[:, :, 40:90, 60:120] = 3
    oh, _ = O.one_hot_mask(mask)
    close(O.assign_identity(oh, None, w).permute(1, 2, 0).reshape(256, 7, 9), golden_ops['id_emb'])


def _run_clip(name, max_frames=None, teacher_forced=False):
    g = np.load(os.path.join(GOLDEN, name))
    former, latter, n, h, wd, oh, ow, gap, objs, seed = g['meta'].tolist()
    frames, mask = make_clip(seed, n, h, wd, objs)
    assert hashlib.sha256(frames.numpy().tobytes()).hexdigest() == str(g['frames_sha'])
    from rmem_ocu_amd.weights import synth_state_dict
    swin = 'swin' in name
    if 'fitted' in name:
        from rmem_ocu_amd.weights import fitted_state_dict
        weights = fitted_state_dict(0)
    else:
        weights = synth_state_dict(0, encoder='swin_base' if swin else 'resnet50', model='deaot' if 'deaot' in name else 'aot')
    if 'deaot' in name:
        from oracle.deaot_cpu import OracleDeAOTEngine
        eng = OracleDeAOTEngine(weights, former, latter, gap)
    else:
        eng = O.OracleEngine(weights, former, latter, gap, align_corners=not swin)
    trace, labels, samples = [], [], []
    eng.long_term_mem_gap = gap
    eng.add_reference_frame(frames[0:1], mask, 0)
    ys, xs = g['sample_y'], g['sample_x']
    inject_at = int(g['inject_at']) if 'inject_at' in g.files else -1
    for i in range(1, n if max_frames is None else min(n, max_frames)):
        logit = eng.match_propogate_one_frame(frames[i:i + 1], (oh, ow))
        label = torch.argmax(torch.softmax(logit, 1), 1, keepdim=True).float()
        own = label
        if teacher_forced:        # the reference's own mask of this frame goes into the memory: every frame is an independent comparison
            label = torch.from_numpy(g['labels'][i - 1].astype(np.float32))[None, None]
        if i == inject_at:        # evaluator.py:484-508
            new = torch.zeros(1, 1, oh, ow)
            new[:, :, oh // 2:oh // 2 + oh // 4, ow // 8:ow // 8 + ow // 5] = objs + 1
            keep = (new == 0).float()
            label = label * keep + new * (1 - keep)
            eng.add_reference_frame(frames[i:i + 1], torch.nn.functional.interpolate(label, size=eng.input_size_2d, mode='nearest'), i)
        else:
            eng.update_memory(torch.nn.functional.interpolate(label, size=eng.input_size_2d, mode='nearest'))
        labels.append((own if teacher_forced else label)[0, 0].to(torch.uint8).numpy())
        trace.append(list(eng.long_memories_indexes))
        samples.append(logit[0][:, ys, xs].numpy())
    return g, np.stack(labels), trace, np.stack(samples)


def test_small_clip_matches_reference():
    """48-frame clip, bank of 3, gap 2: ~20 evictions; masks, eviction trace and logits."""
    g, labels, trace, samples = _run_clip('clip_small.npz')
    width = g['indexes'].shape[1]
    got = -np.ones_like(g['indexes'])
    for i, t in enumerate(trace):
        got[i, :len(t)] = t
    assert (got == g['indexes']).all()
    assert np.abs(samples - g['logit_samples']).max() < 1e-3
    assert (labels == g['labels']).mean() > 0.9999


def _check_clip(name):
    g, labels, trace, samples = _run_clip(name)
    got = -np.ones_like(g['indexes'])
    for i, t in enumerate(trace):
        got[i, :len(t)] = t
    assert (got == g['indexes']).all()
    assert np.abs(samples - g['logit_samples']).max() < 1e-3
    assert (labels == g['labels']).mean() > 0.9999


def test_new_object_clip_matches_reference():
    """cfg-3 protocol: a new object is injected at frame 15 (mid-clip add_reference_frame, quirk 2)."""
    _check_clip('clip_newobj.npz')


def test_unbounded_memory_clip_matches_reference():
    """cfg-4 protocol: latter_mem_len = 9999, the bank grows to 20 entries (temporal-PE slots for T > 4)."""
    _check_clip('clip_unbounded.npz')


def test_restricted_bank_after_injection_raises_like_the_reference():
    """With a small bank the reference raises at the first eviction after a mid-clip reference frame
    (layers/transformer.py:401); the restatement reproduces the failure instead of inventing a behaviour."""
    from rmem_ocu_amd.weights import synth_state_dict
    frames, mask = make_clip(31, 12, 97, 129, 2)
    eng = O.OracleEngine(synth_state_dict(0), 1, 1, 1)
    eng.add_reference_frame(frames[0:1], mask, 0)
    with pytest.raises(RuntimeError):
        for i in range(1, 12):
            logit = eng.match_propogate_one_frame(frames[i:i + 1], (96, 128))
            label = torch.argmax(logit, 1, keepdim=True).float()
            m = torch.nn.functional.interpolate(label, size=eng.input_size_2d, mode='nearest')
            if i == 4:
                eng.add_reference_frame(frames[i:i + 1], m, i)
            else:
                eng.update_memory(m)


def test_swin_encoder_matches_reference():
    """a16: Swin-B stages (window attention with relative-position bias, shifted-window masks, padding, patch merging)."""
    from rmem_ocu_amd.weights import synth_state_dict
    g = np.load(os.path.join(GOLDEN, 'swin_ops.npz'))
    xs = O.encode_image(seeded(2100, (1, 3, 96, 128)), synth_state_dict(0, encoder='swin_base'))
    assert [tuple(x.shape[1:]) for x in xs] == [(128, 24, 32), (256, 12, 16), (512, 6, 8), (256, 6, 8)]
    for i, x in enumerate(xs):
        close(x[0, :, ::2, ::2] if i < 2 else x[0], g[f'swin_x{i}'], 5e-5)


def test_swin_clip_matches_reference():
    """cfg-5 model (SwinB-AOTL, align_corners False, id bank k16 s16) through the engine protocol."""
    _check_clip('clip_swin.npz')


def test_fitted_small_clip_matches_reference():
    """The oracle with the fitted ("trained-like") weights against the reference's free-running clip."""
    _check_clip('clip_small_fitted.npz')


def test_n2_bank_clip_matches_reference():
    """cfg-1 stand-in (BASELINE.json configs[0]: one 82-frame 480p clip, bank N = 2 = 1 + 1, gap 5, one object, fitted weights):
    the oracle against the reference's free-running clip over the first 24 frames (three evictions of the only evictable entry;
    the whole clip is compared on the GPU tier, tests/test_hip_engine.py::test_n2_bank_clip)."""
    g, labels, trace, samples = _run_clip('clip_n2_fitted.npz', max_frames=24)
    k = len(trace)
    got = -np.ones_like(g['indexes'][:k])
    for i, t in enumerate(trace):
        got[i, :len(t)] = t
    assert (got == g['indexes'][:k]).all()
    assert np.abs(samples - g['logit_samples'][:k]).max() < 2e-3
    assert (labels == g['labels'][:k]).mean() > 0.9999


@pytest.mark.parametrize('name', ['clip_long_n8.npz', 'clip_long_n2_fitted.npz'])
def test_long_clip_eviction_traces_match_reference(name):
    """160-frame clips, gap 2 (eviction traces over >= 120 frames): 72 evictions with bank N = 8, 78 with N = 2 --
    the policy's EMA scores and UCB visit counts (layers/transformer.py:357-411) run far past the ~20 evictions of the short
    clips; the oracle is fed the reference's masks (a free run over 160 frames amplifies a single near-tied pixel through the mask feedback,
    which says nothing about either implementation): identical bank index trace at every frame, logits, masks."""
    g, labels, trace, samples = _run_clip(name, teacher_forced=True)
    got = -np.ones_like(g['indexes'])
    for i, t in enumerate(trace):
        got[i, :len(t)] = t
    assert (got == g['indexes']).all()
    evictions = sum(1 for a, b in zip(trace, trace[1:]) if len(a) == len(b) and a != b)
    assert evictions >= 60, evictions
    assert np.abs(samples - g['logit_samples']).max() < 2e-3
    assert (labels == g['labels']).mean() > 0.9999


def test_tta_merge_matches_reference():
    """f3: the test-time-augmentation merge (un-flip, softmax, mean over augmentations, argmax; managers/evaluator.py:427-441)
    against the reference's own flip_tensor + softmax-mean-argmax on seeded logits (tests/golden/tta.npz)."""
    g = np.load(os.path.join(GOLDEN, 'tta.npz'))
    for i in range(int(g['n'])):
        lg = torch.from_numpy(g[f'logits{i}'])
        flips = [bool(f) for f in g[f'flips{i}']]
        prob, label = O.tta_merge([lg[a:a + 1] for a in range(len(flips))], flips)
        assert np.abs(prob.numpy() - g[f'prob{i}']).max() < 1e-6
        assert (label.numpy().astype(np.uint8) == g[f'label{i}']).all()
