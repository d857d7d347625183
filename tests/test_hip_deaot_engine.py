"""End-to-end GPU parity of the DeAOT (R50-DeAOTL) engine against the golden clips the reference produced
(tests/golden/make_golden.py deaot) -- through the drop-in Python API, on the same seeded inputs."""
import numpy as np
import pytest
import torch

from test_hip_engine import _iou, _run, _trace_matrix

pytestmark = pytest.mark.gpu


def test_deaot_small_clip_teacher_forced():
    """48 frames, bank 1 + 2, gap 2: logits within bf16 tolerance, identical eviction trace (the policy's scores and
    visit counts move on every long-term update here, layers/transformer.py:880-968)."""
    g, labels, samples, trace = _run('deaot_clip_small.npz', True)
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('deaot teacher-forced: max |dlogit| =', err, ' logit std =', ref.std(), ' label agreement =', (labels == g['labels']).mean())
    assert err < 0.065 * ref.std(), err      # measured 0.027-0.043 std on MI355X (bf16 stores): 1.5x
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()
    assert (labels == g['labels']).mean() > 0.97


def test_deaot_small_clip_free_running():
    g, labels, samples, trace = _run('deaot_clip_small.npz', False)
    agree = (labels == g['labels']).mean(axis=(1, 2))
    ious = [_iou(a, b) for a, b in zip(g['labels'], labels)]
    print('deaot free-running: label agreement first/last/mean', agree[0], agree[-1], agree.mean(), ' mean IoU', np.mean(ious))
    assert agree[0] > 0.97


def test_deaot_graph_replay_bitwise():
    _, l0, s0, t0 = _run('deaot_clip_small.npz', True, use_graphs=False)
    _, l1, s1, t1 = _run('deaot_clip_small.npz', True, use_graphs=True)
    assert t0 == t1
    assert np.array_equal(s0, s1) and np.array_equal(l0, l1)


def test_deaot_full_clip_cfg2_geometry():
    """481x849 network size (31x54 tokens: window bands, several query tiles), bank 1 + 8 as shipped."""
    g, labels, samples, trace = _run('deaot_clip_full.npz', True, use_graphs=True)
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('deaot full clip: max |dlogit| =', err, ' logit std =', ref.std(), ' agreement =', (labels == g['labels']).mean())
    assert err < 0.065 * ref.std()      # measured <= 0.043 std on MI355X: 1.5x
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()
    assert (labels == g['labels']).mean() > 0.97


def test_deaot_first_frames_vs_oracle():
    """Reference frame + two propagated frames against the CPU oracle on the full logit map (not only sampled pixels)."""
    from oracle.deaot_cpu import OracleDeAOTEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    from test_hip_engine import _engine
    frames, mask = make_clip(5, 4, 161, 193, 3)
    ora = OracleDeAOTEngine(synth_state_dict(0, model='deaot'), 1, 2, 1)
    eng = _engine(1, 2, 1, model_name='r50_deaotl')
    dev = torch.device('cuda', 0)
    ora.add_reference_frame(frames[0:1], mask, 0)
    eng.add_reference_frame(frames[0:1].to(dev), mask.to(dev), obj_nums=[3], frame_step=0)
    for i in range(1, 4):
        lo = ora.match_propogate_one_frame(frames[i:i + 1], (160, 192))
        lg = eng.match_propogate_one_frame(frames[i:i + 1].to(dev), output_size=(160, 192)).cpu()
        err = (lg - lo).abs().max().item()
        print(f'frame {i}: max |dlogit| = {err:.4f}, std = {lo.std().item():.3f}')
        assert err < 0.08 * lo.std().item() + 0.05
        label = torch.argmax(lo, dim=1, keepdim=True).float()
        m = torch.nn.functional.interpolate(label, size=(161, 193), mode='nearest')
        ora.update_memory(m)
        eng.update_memory(m.to(dev))
    assert list(eng.long_memories_indexes) == list(ora.long_memories_indexes)


def test_deaot_small_clip_teacher_forced_fp16():
    """The IEEE-half flavour of the DeAOT path (rmem_gated_attn_f16, rmem_local_gated_attn_f16 and the _f16 GEMMs / norms)."""
    g, labels, samples, trace = _run('deaot_clip_small.npz', True, dtype='fp16')
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('deaot fp16 teacher-forced: max |dlogit| =', err, ' logit std =', ref.std(), ' label agreement =', (labels == g['labels']).mean())
    assert err < 0.012 * ref.std() + 0.003, err
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()


def test_deaot_group_engine_matches_per_clip_engines():
    """Throughput mode of the DeAOT path: three clips in lockstep on one GroupEngine (group_runtime_deaot: batched GEMMs / norms /
    decoder, per-clip gated attentions, scattered bank appends, eviction-policy state moving on every long-term update) deliver the
    masks and eviction traces of three per-clip DeAOT engines."""
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import ClipSlot, GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    from test_hip_engine import _engine
    dev = torch.device('cuda', 0)
    B, n = 3, 26
    clips = [make_clip(60 + c, n, 161, 193, 3) for c in range(B)]
    ref_labels, ref_traces = [], []
    for f, m in clips:
        eng = _engine(1, 2, 5, model_name='r50_deaotl')
        eng.set_async(use_graphs=True)
        slot = ClipSlot(eng, (160, 192), dev, lookahead=4)
        slot.start(f.to(dev), m.to(dev), 3)
        while not slot.done:
            slot.step()
        eng.synchronize()
        ref_labels.append(slot.labels[:n].cpu().numpy().copy())
        ref_traces.append((list(eng.long_memories_indexes), list(eng.aot_engines[0].drop_trace)))
    cfg = get_config('pre_vost', 'test', 'r50_deaotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0, model='deaot'))
    ge = GroupEngine(model, B, 0, 5, lookahead=4)
    gs = GroupSlot(ge, (160, 192), dev)
    gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], 3)
    while not gs.done:
        gs.step()
    ge.synchronize()
    got = gs.labels[:, :n].cpu().numpy()
    for c in range(B):
        agree = (got[c][1:] == ref_labels[c][1:]).mean()
        print(f'deaot clip {c}: label agreement {agree:.5f}, indexes {ge.long_memories_indexes(c)}, drops {ge.drop_trace[c]}')
        assert agree > 0.995
        assert (ge.long_memories_indexes(c), ge.drop_trace[c]) == ref_traces[c]


def test_deaot_new_object_raises_like_the_reference():
    """R50-DeAOTL + a new object mid-clip (managers/evaluator.py:484-508): the REFERENCE raises at the first long-term update after
    the re-added reference frame -- DualBranchGPM.restrict_long_memories has no early return while the bank is not full
    (layers/transformer.py:880-892), so the policy meets a bank of 2 entries and a long_memories_indexes list that kept growing
    (aot_engine.py:322-323) -- observed by running the reference itself (RuntimeError: size of tensor a (2) must match ... (5)).
    The per-clip engine and the clip group (gated attention with a clip dimension, banks of different lengths) reproduce that
    failure at the same update instead of inventing a behaviour; up to there the group delivers the per-clip engine's masks."""
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    from test_hip_engine import _per_clip_reference
    dev = torch.device('cuda', 0)
    B, n, oh, ow, objs = 3, 12, 160, 192, 2
    clips = [make_clip(80 + c, n, 161, 193, objs) for c in range(B)]
    new = torch.zeros(oh, ow, dtype=torch.uint8)
    new[oh // 2:oh // 2 + oh // 4, ow // 8:ow // 8 + ow // 5] = objs + 1
    with pytest.raises(RuntimeError):
        _per_clip_reference(1, 8, 2, clips[1][0], clips[1][1], objs, (oh, ow), new_object=(5, new), model_name='r50_deaotl')
    ref0 = _per_clip_reference(1, 8, 2, clips[0][0][:7], clips[0][1], objs, (oh, ow), model_name='r50_deaotl')
    cfg = get_config('pre_vost', 'test', 'r50_deaotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 8
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0, model='deaot'))
    ge = GroupEngine(model, B, 0, 2, lookahead=2)
    gs = GroupSlot(ge, (oh, ow), dev)
    gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], objs, new_objects={1: (5, new.to(dev))})
    ge.long_term_mem_gap = 2
    steps = 0
    with pytest.raises(RuntimeError):
        while not gs.done:
            gs.step()
            steps += 1
            ge.long_memories_indexes(1)          # resolves the pending policy update of the frame just issued
    ge.synchronize()
    # frame 5 re-adds the reference frame; the long-term update at frame 7 meets a ONE-entry bank (its size-1 score vector broadcasts
    # silently, in the reference too); the one at frame 9 meets two entries against five remembered indexes and fails -- the
    # reference's own message is "The size of tensor a (2) must match the size of tensor b (5)"
    assert 8 <= steps <= 10, steps
    got = gs.labels[0, 1:6].cpu().numpy()
    assert (got == ref0[0][:5]).mean() > 0.995
