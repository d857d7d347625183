"""End-to-end GPU parity of the HIP engine against the golden clips the reference produced
(tests/golden/make_golden.py) -- through the drop-in Python API, on the same seeded inputs."""
import hashlib
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _iou(a, b):
    """mean over object ids of the Jaccard index (evaluation/source/metrics.py:6-37)."""
    ids = [i for i in np.unique(np.concatenate([a.ravel(), b.ravel()])) if i != 0]
    vals = []
    for i in ids:
        u = np.sum((a == i) | (b == i))
        vals.append(1.0 if u == 0 else np.sum((a == i) & (b == i)) / u)
    return float(np.mean(vals)) if vals else 1.0


def _load(name):
    from rmem_ocu_amd.synth import make_clip
    g = np.load(os.path.join(GOLDEN, name))
    former, latter, n, h, w, oh, ow, gap, objs, seed = g['meta'].tolist()
    frames, mask = make_clip(seed, n, h, w, objs)
    assert hashlib.sha256(frames.numpy().tobytes()).hexdigest() == str(g['frames_sha'])
    return g, frames, mask, (former, latter, n, h, w, oh, ow, gap, objs)


def _engine(former, latter, gap, fitted=False, model_name='r50_aotl', dtype='bf16'):
    from rmem_ocu_amd import build_engine, build_vos_model, get_config
    from rmem_ocu_amd.weights import fitted_state_dict, synth_state_dict
    cfg = get_config('pre_vost', 'test', model_name)
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = former, latter
    cfg.MODEL_DTYPE = dtype
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(fitted_state_dict(0) if fitted else synth_state_dict(0, model='deaot' if model_name == 'r50_deaotl' else 'aot'))
    eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=gap)
    eng.eval()
    return eng


def _run(name, teacher_forced, use_graphs=False, dtype='bf16'):
    g, frames, mask, (former, latter, n, h, w, oh, ow, gap, objs) = _load(name)
    inject_at = int(g['inject_at']) if 'inject_at' in g.files else -1
    eng = _engine(former, latter, gap, fitted='fitted' in name, model_name='r50_deaotl' if 'deaot' in name else 'r50_aotl', dtype=dtype)
    eng.use_graphs = use_graphs
    dev = torch.device('cuda', 0)
    frames_d = frames.to(dev)
    eng.add_reference_frame(frames_d[0:1], mask.to(dev), obj_nums=[int(mask.max())], frame_step=0)
    ys, xs = torch.from_numpy(g['sample_y']).to(dev), torch.from_numpy(g['sample_x']).to(dev)
    labels, samples, trace = [], [], []
    for i in range(1, n):
        logit = eng.match_propogate_one_frame(frames_d[i:i + 1], output_size=(oh, ow))
        prob = torch.softmax(logit, dim=1)
        label = torch.argmax(prob, dim=1, keepdim=True).float()
        fed = torch.from_numpy(g['labels'][i - 1].astype(np.float32)).to(dev)[None, None] if teacher_forced else label
        if i == inject_at:        # evaluator.py:484-508 (the golden label of this frame already contains the new object)
            if not teacher_forced:
                new = torch.zeros(1, 1, oh, ow, device=dev)
                new[:, :, oh // 2:oh // 2 + oh // 4, ow // 8:ow // 8 + ow // 5] = objs + 1
                fed = label * (new == 0).float() + new
            eng.add_reference_frame(frames_d[i:i + 1], F.interpolate(fed, size=eng.input_size_2d, mode='nearest'),
                                    obj_nums=[int(fed.max().item())], frame_step=i)
            label = fed
        else:
            eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode='nearest'))
        labels.append(label[0, 0].to(torch.uint8).cpu().numpy())
        samples.append(logit[0][:, ys, xs].cpu().numpy())
        trace.append(list(eng.long_memories_indexes))
    return g, np.stack(labels), np.stack(samples), trace


def _trace_matrix(trace, like):
    got = -np.ones_like(like)
    for i, t in enumerate(trace):
        got[i, :len(t)] = t
    return got


def test_small_clip_teacher_forced():
    """Per-frame parity with the reference's masks fed back (no error feedback): logits within bf16
    tolerance, identical eviction trace."""
    g, labels, samples, trace = _run('clip_small.npz', True)
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('teacher-forced: max |dlogit| =', err, ' logit std =', ref.std(), ' label agreement =', (labels == g['labels']).mean())
    assert err < 0.065 * ref.std(), err      # measured 0.027-0.043 std on MI355X (bf16 stores): 1.5x
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()
    assert (labels == g['labels']).mean() > 0.97


def test_small_clip_free_running():
    g, labels, samples, trace = _run('clip_small.npz', False)
    agree = (labels == g['labels']).mean(axis=(1, 2))
    ious = [_iou(a, b) for a, b in zip(g['labels'], labels)]
    print('free-running: label agreement first/last/mean', agree[0], agree[-1], agree.mean(), ' mean IoU', np.mean(ious))
    assert agree[0] > 0.97
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()


def test_small_clip_graph_replay_bitwise():
    """hipGraph replay of the frame launch lists gives bit-identical logits to direct launches."""
    _, l0, s0, t0 = _run('clip_small.npz', True, use_graphs=False)
    _, l1, s1, t1 = _run('clip_small.npz', True, use_graphs=True)
    assert t0 == t1
    assert np.array_equal(s0, s1) and np.array_equal(l0, l1)


def test_full_clip_cfg2_geometry():
    """481x849 network size, bank N = 8: masks / eviction trace against the reference's golden clip."""
    if not os.path.exists(os.path.join(GOLDEN, 'clip_full.npz')):
        pytest.skip('clip_full.npz not generated')
    g, labels, samples, trace = _run('clip_full.npz', True)
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('full clip teacher-forced: max |dlogit| =', err, ' logit std =', ref.std(), ' agreement =', (labels == g['labels']).mean())
    assert err < 0.065 * ref.std()      # measured <= 0.043 std on MI355X: 1.5x
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()


def test_new_object_injection_clip():
    """cfg-3 protocol (quirk 2): mid-clip add_reference_frame re-initialises the bank, indexes keep growing."""
    g, labels, samples, trace = _run('clip_newobj.npz', True)
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('new-object clip: max |dlogit| =', err, ' agreement =', (labels == g['labels']).mean())
    assert err < 0.065 * ref.std()      # measured <= 0.043 std on MI355X: 1.5x
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()


def test_unbounded_memory_clip():
    """cfg-4 protocol: the bank grows past its initial ring (grow_bank), T up to 20, chunk table of 20 frames."""
    g, labels, samples, trace = _run('clip_unbounded.npz', True, use_graphs=True)
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('unbounded clip: max |dlogit| =', err, ' agreement =', (labels == g['labels']).mean())
    assert err < 0.065 * ref.std()      # measured <= 0.043 std on MI355X: 1.5x
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()


def test_more_than_ten_objects_vs_oracle():
    """12 objects -> two engines (separate_mask + soft_logit_aggregation, aot_engine.py:604-673).  The reference itself
    cannot run this case (its engines share one LSTT memory and it raises at the first eviction), so the checker is the
    oracle's per-engine-state restatement: parity for this row is NOT pinned by a reference fixture."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    frames, mask = make_clip(51, 8, 161, 193, 12)
    assert int(mask.max()) == 12
    dev = torch.device('cuda', 0)
    eng = _engine(1, 2, 2)
    ora = O.OracleInferEngine(synth_state_dict(0), 1, 2, 2)
    eng.add_reference_frame(frames[0:1].to(dev), mask.to(dev), obj_nums=[12], frame_step=0)
    ora.add_reference_frame(frames[0:1], mask, 12, 0)
    assert len(eng.aot_engines) == 2
    worst, agree = 0.0, []
    for i in range(1, 8):
        got = eng.match_propogate_one_frame(frames[i:i + 1].to(dev), output_size=(160, 192)).cpu()
        ref = ora.match_propogate_one_frame(frames[i:i + 1], (160, 192))
        assert got.shape == ref.shape == (1, 21, 160, 192)
        pg, pr = torch.softmax(got, 1), torch.softmax(ref, 1)
        worst = max(worst, (pg - pr).abs().max().item())
        label = torch.argmax(pr, dim=1, keepdim=True).float()
        agree.append((pg - pr).abs().mean().item())       # 21 near-tied classes with synthetic weights: compare probabilities, not argmax
        m = F.interpolate(label, size=(161, 193), mode='nearest')
        eng.update_memory(m.to(dev))
        ora.update_memory(m)
    print('12 objects: max |dprob| =', worst, ' mean |dprob| =', np.mean(agree), '(uniform = %.3f)' % (1 / 21))
    assert worst < 0.01 and np.mean(agree) < 0.001
    assert eng.long_memories_indexes == ora.long_memories_indexes


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_swin_encoder_and_clip(dtype):
    """cfg-5 model on the GPU: Swin-B stage outputs against the reference fixture, then the SwinB-AOTL clip
    (align_corners False, id bank k16 s16) teacher-forced against the reference's golden clip.  BASELINE cfg 5 names fp16:
    the 'fp16' case runs every kernel through its _f16 entry point and is held to 8x tighter bounds."""
    tol_stage, tol_logit = (0.04, 0.045) if dtype == 'bf16' else (0.006, 0.008)
    from rmem_ocu_amd import build_engine, build_vos_model, get_config, ops
    from rmem_ocu_amd.runtime import ClipRuntime
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    cfg = get_config('pre_vost', 'test', 'swinb_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    cfg.MODEL_DTYPE = dtype
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    sd = synth_state_dict(0, encoder='swin_base')
    model.load_state_dict(sd)
    # --- encoder stages
    g = np.load(os.path.join(GOLDEN, 'swin_ops.npz'))
    rng = np.random.Generator(np.random.PCG64([2100, 0xC0FFEE]))
    img = torch.from_numpy(rng.standard_normal((1, 3, 96, 128)).astype(np.float32))
    rt = ClipRuntime(model.packed(), (96, 128), 4, dev, 3, False, 11)
    imgd = img[0].to(dev).contiguous()
    ops.run(rt.prog_encode(imgd))
    torch.cuda.synchronize()
    for i, (buf, (h, w, c)) in enumerate(zip((rt.enc1, rt.enc2, rt.enc3), ((24, 32, 128), (12, 16, 256), (6, 8, 512)))):
        got = buf.view(-1)[: h * w * c].float().view(h, w, c).permute(2, 0, 1).cpu().numpy()
        ref = g[f'swin_x{i}']
        got = got[:, ::2, ::2] if i < 2 else got
        err = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-6)
        print(f'swin stage {i} [{dtype}]: rel err {err:.4f}')
        assert err < tol_stage, (i, err)
    # --- clip
    gc, frames, mask, (former, latter, n, h, w, oh, ow, gap, objs) = _load('clip_swin.npz')
    eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=gap)
    fd = frames.to(dev)
    eng.add_reference_frame(fd[0:1], mask.to(dev), obj_nums=[objs], frame_step=0)
    ys, xs = torch.from_numpy(gc['sample_y']).to(dev), torch.from_numpy(gc['sample_x']).to(dev)
    samples, trace = [], []
    for i in range(1, n):
        logit = eng.match_propogate_one_frame(fd[i:i + 1], output_size=(oh, ow))
        fed = torch.from_numpy(gc['labels'][i - 1].astype(np.float32)).to(dev)[None, None]
        eng.update_memory(F.interpolate(fed, size=eng.input_size_2d, mode='nearest'))
        samples.append(logit[0][:, ys, xs].cpu().numpy())
        trace.append(list(eng.long_memories_indexes))
    ref = gc['logit_samples']
    err = np.abs(np.stack(samples) - ref).max()
    print(f'swin clip [{dtype}]: max |dlogit| =', err, ' logit std =', ref.std())
    assert err < tol_logit * ref.std()      # measured 0.027 std in bf16
    assert (_trace_matrix(trace, gc['indexes']) == gc['indexes']).all()


@pytest.mark.parametrize('name', ['clip_small_fitted.npz', 'clip_full_fitted.npz'])
def test_fitted_weights_mask_iou(name):
    """Mask parity with the fitted ("trained-like") weights, where the reference's masks are confident.
    Per-frame (teacher-forced: the reference's mask of frame i-1 is fed back, so every frame is an independent comparison of
    the same computation): mean IoU over object ids and frames must be >= 0.99 and the eviction trace identical.
    Free-running (own masks fed back for the whole clip) is reported and loosely bounded: with these weights the reference
    itself tracks the synthetic objects poorly (80 % agreement with ground truth), so small bf16 differences at mask
    borders are amplified frame over frame; the north-star 0.999 figure is for trained DAVIS weights (unavailable offline)."""
    if not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('tests/golden/trained_delta.pt missing (python tests/golden/train_synth_weights.py)')
    g, labels, samples, trace = _run(name, True, use_graphs=True)
    ious = [_iou(a, b) for a, b in zip(g['labels'], labels)]
    agree = (labels == g['labels']).mean(axis=(1, 2))
    ref = g['logit_samples']
    print(f'{name} teacher-forced: mean IoU {np.mean(ious):.5f} min IoU {np.min(ious):.5f}  label agreement {agree.mean():.5f}  '
          f'max |dlogit| {np.abs(samples - ref).max():.3f} at logit std {ref.std():.2f}')
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()
    assert np.mean(ious) >= 0.99 and agree.mean() >= 0.999
    assert np.abs(samples - ref).max() < 0.035 * ref.std()      # measured 0.019-0.023 std (bf16: 8 significant bits)
    g, labels, samples, trace = _run(name, False, use_graphs=True)
    ious = [_iou(a, b) for a, b in zip(g['labels'], labels)]
    agree = (labels == g['labels']).mean(axis=(1, 2))
    print(f'{name} free-running: mean IoU {np.mean(ious):.5f} first-frame IoU {ious[0]:.5f} last-frame IoU {ious[-1]:.5f}  '
          f'label agreement {agree.mean():.5f}')
    assert ious[0] >= 0.985 and agree.mean() >= 0.9


@pytest.mark.parametrize('name', ['clip_small_fitted.npz', 'clip_full_fitted.npz'])
def test_fitted_weights_mask_iou_fp16(name):
    """The north-star gate (>= 0.999 mask IoU against the reference's masks) in the IEEE-half flavour (cfg.MODEL_DTYPE = 'fp16':
    the operand type of the reference's own --amp path, tools/eval.py:45-47).  bfloat16's 8 significant bits cap the same
    comparison at 0.992-0.996 (test_fitted_weights_mask_iou; tests/test_stage_budget.py shows the ~1 % rms the 50 bf16-stored
    encoder layers accumulate); half's 11 bits bring every stage's error down 8x.  Measured on MI355X: mean IoU 0.9993 on the
    161x193 clip and 0.9988 on the cfg-2-size clip (0.0078 % of the pixels flip, ~30 per frame, all on object borders), so the
    gate is 0.999 / 0.998 -- 1.5x the measured shortfall, not the 0.99 of the bfloat16 flavour."""
    if not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('fitted weights missing')
    g, labels, samples, trace = _run(name, True, use_graphs=True, dtype='fp16')
    ious = [_iou(a, b) for a, b in zip(g['labels'], labels)]
    agree = (labels == g['labels']).mean(axis=(1, 2))
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print(f'{name} fp16 teacher-forced: mean IoU {np.mean(ious):.5f} min IoU {np.min(ious):.5f}  label agreement {agree.mean():.5f}  '
          f'max |dlogit| {err:.4f} at logit std {ref.std():.2f}')
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()
    assert np.mean(ious) >= (0.999 if 'small' in name else 0.998) and agree.mean() >= 0.9998
    assert err < 0.006 * ref.std() + 0.005, err


def test_small_clip_teacher_forced_fp16():
    """Synthetic weights, IEEE-half flavour: per-frame logits against the reference's, identical eviction trace."""
    g, labels, samples, trace = _run('clip_small.npz', True, dtype='fp16')
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print('fp16 teacher-forced: max |dlogit| =', err, ' logit std =', ref.std(), ' label agreement =', (labels == g['labels']).mean())
    assert err < 0.012 * ref.std() + 0.003, err
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_n2_bank_clip(dtype):
    """BASELINE cfg 1 stand-in on the HIP path: 82 frames at 481x849, bank N = 2 (1 + 1), gap 5, one object, fitted weights --
    teacher-forced against the reference's golden clip: identical bank trace through all 15 evictions, per-frame mask IoU."""
    if not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('fitted weights missing')
    g, labels, samples, trace = _run('clip_n2_fitted.npz', True, use_graphs=True, dtype=dtype)
    ious = [_iou(a, b) for a, b in zip(g['labels'], labels)]
    ref = g['logit_samples']
    err = np.abs(samples - ref).max()
    print(f'N=2 clip [{dtype}]: mean IoU {np.mean(ious):.5f} min IoU {np.min(ious):.5f}  max |dlogit| {err:.4f} at logit std {ref.std():.2f}')
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()
    assert np.mean(ious) >= (0.99 if dtype == 'bf16' else 0.998)
    assert err < (0.035 if dtype == 'bf16' else 0.006) * ref.std() + 0.005


def test_full_geometry_cfg3_cfg4_engine():
    """BASELINE cfg 3 / 4 geometry on the engine: 720x1280 video -> network size 577x1041 (HW = 37 x 66 = 2442 tokens), unbounded
    bank (latter_mem_len = 9999) grown with gap 1 to T = 30 entries (cfg 4 reaches 30).  No golden clip exists at this size, so:
    the first 4 propagated frames are compared with the fp32 oracle (teacher-forced with the oracle's labels), and the whole
    31-frame run replayed as hipGraphs must be bit-identical to direct launches, with the bank trace 0..29."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd.synth import make_clip, network_size
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    vh, vw = 720, 1280
    h, w = network_size(vh, vw)
    assert (h, w) == (577, 1041)
    n, objs = 31, 2
    frames, mask = make_clip(301, n, h, w, objs)
    fd = frames.to(dev)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ora = O.OracleEngine(synth_state_dict(0), 1, 9999, 1)
    runs = []
    for use_graphs in (False, True):
        eng = _engine(1, 9999, 1)
        eng.use_graphs = use_graphs
        eng.add_reference_frame(fd[0:1], mask.to(dev), obj_nums=[objs], frame_step=0)
        if not use_graphs:
            ora.add_reference_frame(frames[0:1], mask, 0)
        outs = []
        for i in range(1, n):
            logit = eng.match_propogate_one_frame(fd[i:i + 1], output_size=(vh, vw))
            if not use_graphs and i <= 4:
                with torch.no_grad():
                    ref = ora.match_propogate_one_frame(frames[i:i + 1], (vh, vw))
                err = (logit.cpu() - ref).abs().max().item() / ref.std().item()
                print(f'577x1041 frame {i}: max |dlogit| / std = {err:.4f} (bank T = {len(ora.long_memories_indexes)})')
                assert err < 0.065
                lab = torch.argmax(ref, dim=1, keepdim=True).float()
                ora.update_memory(F.interpolate(lab, size=(h, w), mode='nearest'))
            else:
                lab = torch.argmax(logit, dim=1, keepdim=True).float().cpu()
            eng.update_memory(F.interpolate(lab, size=(h, w), mode='nearest').to(dev))
            outs.append(logit[0, :, ::16, ::16].cpu().numpy().copy())
        runs.append((np.stack(outs), list(eng.long_memories_indexes)))
    assert runs[0][1] == list(range(31)) == runs[1][1]          # 31 entries after the last update; the last propagation read T = 30
    # frames 1..4 of the two runs were fed different labels (oracle vs own argmax) only if they disagree; compare from the
    # point where both runs are self-fed and therefore must be the same computation: the graph run is self-fed throughout, so it
    # is compared with a second self-fed direct run instead
    eng = _engine(1, 9999, 1)
    eng.add_reference_frame(fd[0:1], mask.to(dev), obj_nums=[objs], frame_step=0)
    outs = []
    for i in range(1, n):
        logit = eng.match_propogate_one_frame(fd[i:i + 1], output_size=(vh, vw))
        lab = torch.argmax(logit, dim=1, keepdim=True).float()
        eng.update_memory(F.interpolate(lab, size=(h, w), mode='nearest'))
        outs.append(logit[0, :, ::16, ::16].cpu().numpy().copy())
    assert np.array_equal(np.stack(outs), runs[1][0]), 'hipGraph replay differs from direct launches at HW = 2442, T -> 30'


def test_group_engine_cfg3_geometry_new_object_vs_oracle():
    """BASELINE cfg 3 protocol at cfg 3 / 4 GEOMETRY on the throughput path: 720x1280 video -> network size 577x1041 (HW = 2442), restricted
    bank N = 8 (1 + 7), two clips in lockstep on one GroupEngine (hipGraphs, look-ahead encoder, chain kernels), a new object's mask
    arriving at frame 5 of clip 1 only (managers/evaluator.py:484-508).  No reference fixture exists at this size, so the fp32 oracle runs
    both clips (shared prefix, then with / without the new object) and its labels are fed back: logits, labels and the bank index trace
    of every frame and clip against the oracle, through bank fill, four evictions (gap 1) and the restarted bank of clip 1 (kept below
    N: at the first eviction after a re-added reference frame the reference itself raises, test_oracle_golden.py)."""
    import copy
    from oracle import ref_cpu as O
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip, network_size
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    vh, vw = 720, 1280
    h, w = network_size(vh, vw)
    n, objs, inj = 12, 2, 5        # clip 1 ends with 7 entries: the reference raises at the first eviction after a re-added reference frame
    frames, mask = make_clip(303, n, h, w, objs)
    new = torch.zeros(vh, vw, dtype=torch.uint8)
    new[vh // 2:vh // 2 + vh // 4, vw // 8:vw // 8 + vw // 5] = objs + 1
    torch.set_num_threads(min(16, os.cpu_count() or 1))

    def oracle_frames(eng, lo, inject):
        out = []
        with torch.no_grad():
            for i in range(lo, n):
                logit = eng.match_propogate_one_frame(frames[i:i + 1], (vh, vw))
                own = torch.argmax(logit, dim=1, keepdim=True).float()
                label = own
                if inject and i == inj:
                    label = torch.where(new[None, None] > 0, new[None, None].float(), own)
                    eng.add_reference_frame(frames[i:i + 1], F.interpolate(label, size=(h, w), mode='nearest'), i)
                else:
                    eng.update_memory(F.interpolate(label, size=(h, w), mode='nearest'))
                out.append((own[0, 0].to(torch.uint8), logit[0, :, ::16, ::16].clone(), list(eng.long_memories_indexes)))
                if i + 1 == inj and not inject:
                    return out
        return out

    ora = O.OracleEngine(synth_state_dict(0), 1, 7, 1)
    ora.long_term_mem_gap = 1
    ora.add_reference_frame(frames[0:1], mask, 0)
    prefix = oracle_frames(ora, 1, False)                     # frames 1 .. inj - 1, common to both clips
    assert len(prefix) == inj - 1
    refs = [prefix + oracle_frames(copy.deepcopy(ora), inj, False), prefix + oracle_frames(ora, inj, True)]

    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 7
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    ge = GroupEngine(model, 2, 0, 1, lookahead=2)
    assert ge.use_graphs
    gs = GroupSlot(ge, (vh, vw), dev)
    fd = frames.to(dev)
    gs.start([fd, fd], [mask.to(dev)] * 2, objs, new_objects={1: (inj, new.to(dev))})
    ge.long_term_mem_gap = 1
    gold = torch.stack([torch.stack([refs[c][i][0] for c in range(2)]) for i in range(n - 1)]).to(dev)   # [n - 1, 2, vh, vw]
    torch.cuda.synchronize()
    rt = ge.rt
    worst, banks = 0.0, []
    while not gs.done:
        i = gs.cursor
        gs.step(feed=gold[i - 1])
        ge.synchronize()
        lg = rt.logits.view(2, rt.H4, rt.W4, 16)[..., :11].permute(0, 3, 1, 2)
        up = F.interpolate(lg, size=(vh, vw), mode='bilinear', align_corners=True)[:, :, ::16, ::16].cpu()
        banks.append([len(sl) for sl in rt.slots])
        for c in range(2):
            own, ref, trace = refs[c][i - 1]
            err = (up[c] - ref).abs().max().item() / ref.std().item()
            worst = max(worst, err)
            agree = (gs.labels[c, i].cpu() == own).float().mean().item()
            assert list(ge.long_memories_indexes(c)) == trace, (c, i, list(ge.long_memories_indexes(c)), trace)
            assert err < 0.065 and agree > 0.97, (c, i, err, agree)
    print(f'577x1041 group, N = 8, new object at frame {inj} of clip 1: worst max |dlogit| / std {worst:.4f}; bank sizes per frame {banks}')
    # clip 0 filled its bank at frame 7 and evicted since (a slot list holds 9 entries while an eviction is pending); clip 1 restarted at frame 5
    assert banks[inj - 1] == [inj + 1, 1] and banks[-1][0] in (8, 9) and banks[-1][1] == n - inj and len(refs[0][-1][2]) == 8


def test_full_geometry_cfg5_swin_engine():
    """BASELINE cfg 5 geometry on the engine: Swin-B at 720x1280 (HW = 45 x 80 = 3600 tokens), bank N = 12 (1 + 11), fp16 as the
    config names it: 16 frames with gap 1 fill the bank and evict; graph replay bit-identical to direct launches, logits finite,
    bank never above 12 entries; the first three frames against the fp32 oracle at this size."""
    from rmem_ocu_amd import build_engine, build_vos_model, get_config
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    h, w, n = 720, 1280, 16
    frames, mask = make_clip(302, n, h, w, 2)
    fd = frames.to(dev)
    cfg = get_config('pre_vost', 'test', 'swinb_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 11
    cfg.MODEL_DTYPE = 'fp16'
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0, encoder='swin_base'))
    runs = []
    for use_graphs in (False, True):
        eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=1)
        eng.use_graphs = use_graphs
        eng.add_reference_frame(fd[0:1], mask.to(dev), obj_nums=[2], frame_step=0)
        outs, sizes = [], []
        for i in range(1, n):
            logit = eng.match_propogate_one_frame(fd[i:i + 1], output_size=(h, w))
            assert torch.isfinite(logit).all()
            lab = torch.argmax(logit, dim=1, keepdim=True).float()
            eng.update_memory(F.interpolate(lab, size=eng.input_size_2d, mode='nearest'))
            outs.append(logit[0, :, ::16, ::16].cpu().numpy().copy())
            sizes.append(len(eng.long_memories_indexes))
        runs.append((np.stack(outs), sizes))
    assert max(runs[0][1]) == 12 and runs[0][1] == runs[1][1]
    assert np.array_equal(runs[0][0], runs[1][0]), 'hipGraph replay differs from direct launches at HW = 3600, N = 12'
    # and against the fp32 oracle at this size (no reference fixture exists at 720 x 1280): the first three propagated frames,
    # teacher-forced with the oracle's labels, logits at the fp16 tolerance of the small Swin clip
    from oracle import ref_cpu as O
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ora = O.OracleEngine(synth_state_dict(0, encoder='swin_base'), 1, 11, 1, align_corners=False)
    eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=1)
    eng.add_reference_frame(fd[0:1], mask.to(dev), obj_nums=[2], frame_step=0)
    with torch.no_grad():
        ora.add_reference_frame(frames[0:1], mask, 0)
        for i in range(1, 4):
            logit = eng.match_propogate_one_frame(fd[i:i + 1], output_size=(h, w))
            ref = ora.match_propogate_one_frame(frames[i:i + 1], (h, w))
            err = (logit.cpu() - ref).abs().max().item() / ref.std().item()
            agree = (torch.argmax(logit.cpu(), 1) == torch.argmax(ref, 1)).float().mean().item()
            print(f'720x1280 Swin-B fp16 frame {i}: max |dlogit| / std = {err:.4f}, label agreement {agree:.5f} (bank T = {len(ora.long_memories_indexes)})')
            assert err < 0.009 and agree > 0.998          # 1.5 x the 0.0055 / 0.99926 measured on MI355X
            lab = torch.argmax(ref, dim=1, keepdim=True).float()
            ora.update_memory(F.interpolate(lab, size=ora.input_size_2d, mode='nearest'))
            eng.update_memory(F.interpolate(lab, size=eng.input_size_2d, mode='nearest').to(dev))
    assert list(eng.long_memories_indexes) == list(ora.long_memories_indexes)


def test_sequence_evaluator_flip_tta_and_metrics(tmp_path):
    """f3 / f4: the evaluator protocol with horizontal-flip TTA (two engines, probabilities averaged on the device), J per
    object from the device counts, palette PNG output -- against the oracle's evaluate_sequence on the fitted weights."""
    if not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('fitted weights missing')
    from oracle import ref_cpu as O
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.evaluator import SequenceEvaluator, region_similarity, save_mask
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import fitted_state_dict
    dev = torch.device('cuda', 0)
    sd = fitted_state_dict(0)
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(sd)
    frames, mask = make_clip(71, 7, 161, 193, 2)
    first = F.interpolate(mask.float(), size=(160, 192), mode='nearest')
    ref_labels, ref_probs = O.evaluate_sequence(sd, frames, {0: first}, (160, 192), 1, 2, flip=True)
    ev = SequenceEvaluator(model, 0, flip=True)
    got = ev.run(frames.to(dev), {0: first.to(dev)}, (160, 192))
    assert len(got) == len(ref_labels) == 6 and len(ev.engines) == 2
    agree = [(g.cpu() == r).float().mean().item() for g, r in zip(got, ref_labels)]
    print('flip-TTA sequence: label agreement per frame', [round(a, 4) for a in agree])
    assert agree[0] > 0.997 and min(agree) > 0.97
    # J from the device counts equals the numpy metric
    j = region_similarity(got[0], ref_labels[0].to(dev))
    for i, v in j.items():
        assert abs(v - O.db_eval_iou((ref_labels[0].numpy() == i), (got[0].cpu().numpy() == i))) < 1e-9
    assert min(j.values()) > 0.98
    # palette PNG round trip
    p = str(tmp_path / 'm.png')
    save_mask(got[0].cpu().numpy(), p)
    from PIL import Image
    im = Image.open(p)
    assert im.mode == 'P' and np.array_equal(np.array(im), got[0].cpu().numpy())
    assert im.getpalette()[:9] == [0, 0, 0, 128, 0, 0, 0, 128, 0]


def test_encoder_lookahead_matches_per_frame_path():
    """ClipSlot with the ResNet-50 encoder running 4 frames ahead (one launch per layer for 4 frames) delivers the same
    masks as the frame-by-frame path (split-K plans differ with the GEMM's row count, so logits agree to bf16 noise)."""
    from rmem_ocu_amd.clip_runner import ClipSlot
    from rmem_ocu_amd.synth import make_clip
    dev = torch.device('cuda', 0)
    frames, mask = make_clip(7, 14, 161, 193, 3)
    out = []
    for la in (1, 4):
        eng = _engine(1, 2, 2)
        eng.set_async(use_graphs=True)
        slot = ClipSlot(eng, (160, 192), dev, lookahead=la)
        slot.start(frames.to(dev), mask.to(dev), 3)
        while not slot.done:
            slot.step()
        eng.synchronize()
        out.append((slot.labels[:14].cpu().numpy().copy(), list(eng.long_memories_indexes)))
    (l1, t1), (l4, t4) = out
    agree = (l1[1:] == l4[1:]).mean()
    print('look-ahead 4 vs 1: label agreement', agree)
    assert t1 == t4
    assert agree > 0.995


def test_swin_encoder_lookahead_matches_per_frame_path():
    """The Swin-B encoder running 3 frames ahead (encoder_batch.SwinBatchEncoder: every linear / LayerNorm one launch over the
    rows of 3 frames, window attention and patch merging per frame) against the frame-by-frame path: same masks, same bank trace.
    The network size is NOT a multiple of the 7-token window (the padded / shifted windows of every frame must stay its own)."""
    from rmem_ocu_amd import build_engine, build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import ClipSlot
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    frames, mask = make_clip(9, 11, 160, 192, 3)
    cfg = get_config('pre_vost', 'test', 'swinb_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    cfg.MODEL_DTYPE = 'fp16'
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0, encoder='swin_base'))
    out = []
    for la in (1, 3):
        eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=2)
        eng.eval()
        eng.set_async(use_graphs=True)
        slot = ClipSlot(eng, (160, 192), dev, lookahead=la)
        slot.start(frames.to(dev), mask.to(dev), 3)
        while not slot.done:
            slot.step()
        eng.synchronize()
        out.append((slot.labels[:11].cpu().numpy().copy(), list(eng.long_memories_indexes)))
    (l1, t1), (l3, t3) = out
    agree = (l1[1:] == l3[1:]).mean()
    print('swin look-ahead 3 vs 1: label agreement', agree)
    assert t1 == t3
    assert agree > 0.999


def test_swin_group_engine_matches_per_clip_engines():
    """Throughput mode of the cfg-5 model: three clips in lockstep on one GroupEngine over SwinB-AOTL (Swin-B look-ahead encoder
    over 3 x 2 frames, align_corners False, 16x16 stride-16 identity bank, fp16) against three per-clip engines."""
    from rmem_ocu_amd import build_engine, build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import ClipSlot, GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    B, n = 3, 16
    clips = [make_clip(80 + c, n, 160, 192, 3) for c in range(B)]
    cfg = get_config('pre_vost', 'test', 'swinb_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    cfg.MODEL_DTYPE = 'fp16'
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0, encoder='swin_base'))
    ref_labels, ref_traces = [], []
    for f, m in clips:
        eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=5)
        eng.eval()
        eng.set_async(use_graphs=True)
        slot = ClipSlot(eng, (160, 192), dev, lookahead=1)
        slot.start(f.to(dev), m.to(dev), 3)
        while not slot.done:
            slot.step()
        eng.synchronize()
        ref_labels.append(slot.labels[:n].cpu().numpy().copy())
        ref_traces.append((list(eng.long_memories_indexes), list(eng.aot_engines[0].drop_trace)))
    ge = GroupEngine(model, B, 0, 5, lookahead=2)
    gs = GroupSlot(ge, (160, 192), dev)
    gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], 3)
    while not gs.done:
        gs.step()
    ge.synchronize()
    got = gs.labels[:, :n].cpu().numpy()
    for c in range(B):
        agree = (got[c][1:] == ref_labels[c][1:]).mean()
        print(f'swin clip {c}: label agreement {agree:.5f}, indexes {ge.long_memories_indexes(c)}, drops {ge.drop_trace[c]}')
        assert agree > 0.998
        assert (ge.long_memories_indexes(c), ge.drop_trace[c]) == ref_traces[c]


def test_sequence_evaluator_multiscale_tta():
    """f3: multi-scale + flip testing (TEST_MULTISCALE = [1.0, 1.3], TEST_FLIP): four engines at two network sizes, logits
    resized to the original size and averaged (managers/evaluator.py:342-355, 427-438) -- against the oracle."""
    if not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('fitted weights missing')
    from oracle import ref_cpu as O
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.evaluator import SequenceEvaluator
    from rmem_ocu_amd.synth import make_clip, network_size
    from rmem_ocu_amd.weights import fitted_state_dict
    dev = torch.device('cuda', 0)
    sd = fitted_state_dict(0)
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(sd)
    out_hw = (160, 192)
    assert network_size(*out_hw) == (161, 193) and network_size(*out_hw, scale=1.3) == (209, 257)
    frames, mask = make_clip(72, 6, 161, 193, 2)
    big = F.interpolate(frames, size=(209, 257), mode='bilinear', align_corners=True)    # stand-in for the loader's resize
    first = F.interpolate(mask.float(), size=out_hw, mode='nearest')
    ref_labels, _ = O.evaluate_sequence(sd, [frames, big], {0: first}, out_hw, 1, 2, flip=True)
    ev = SequenceEvaluator(model, 0, flip=True)
    got = ev.run([frames.to(dev), big.to(dev)], {0: first.to(dev)}, out_hw)
    assert len(ev.engines) == 4 and len(got) == 5
    agree = [(g.cpu() == r).float().mean().item() for g, r in zip(got, ref_labels)]
    print('multi-scale + flip TTA: label agreement per frame', [round(a, 4) for a in agree])
    assert agree[0] > 0.997 and min(agree) > 0.97


def test_clip_slot_from_pinned_uint8_frames():
    """PCIe-inclusive path: decoded uint8 frames in pinned host memory -> H2D -> rmem_ingest_rgb8 -> look-ahead encoder gives
    the same masks as feeding the ingested fp32 frames from device memory."""
    from rmem_ocu_amd import ops
    from rmem_ocu_amd.clip_runner import ClipSlot
    from rmem_ocu_amd.synth import make_clip
    dev = torch.device('cuda', 0)
    frames, mask = make_clip(9, 11, 161, 193, 2)
    vid = F.interpolate(frames, size=(160, 192), mode='bilinear', align_corners=False)
    u8 = (vid * 40.0 + 128.0).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().pin_memory()
    u8d = u8.to(dev)
    ing = torch.empty(11, 3, 161, 193, dtype=torch.float32, device=dev)
    ops.run([ops.ingest_rgb8(u8d[i], Hs=160, Ws=192, Hd=161, Wd=193, out_chw=ing[i]) for i in range(11)])
    torch.cuda.synchronize()
    out = []
    for src in (ing, u8):
        eng = _engine(1, 2, 2)
        eng.set_async(use_graphs=True)
        slot = ClipSlot(eng, (160, 192), dev, lookahead=4)
        slot.start(src, mask.to(dev), 2)
        while not slot.done:
            slot.step()
        eng.synchronize()
        out.append(slot.labels[:11].cpu().numpy().copy())
    assert np.array_equal(out[0][1:], out[1][1:])


def test_group_engine_matches_per_clip_engines():
    """Three clips in lockstep on one GroupEngine (one launch per layer for the group, per-clip banks / eviction policies) deliver
    the masks and eviction traces of three per-clip engines."""
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import ClipSlot, GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    B, n = 3, 26
    clips = [make_clip(40 + c, n, 161, 193, 3) for c in range(B)]
    ref_labels, ref_traces = [], []
    for f, m in clips:
        eng = _engine(1, 2, 5)
        eng.set_async(use_graphs=True)
        slot = ClipSlot(eng, (160, 192), dev, lookahead=4)
        slot.start(f.to(dev), m.to(dev), 3)
        while not slot.done:
            slot.step()
        eng.synchronize()
        ref_labels.append(slot.labels[:n].cpu().numpy().copy())
        ref_traces.append((list(eng.long_memories_indexes), list(eng.aot_engines[0].drop_trace)))
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    ge = GroupEngine(model, B, 0, 5, lookahead=4)
    gs = GroupSlot(ge, (160, 192), dev)
    gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], 3)
    while not gs.done:
        gs.step()
    ge.synchronize()
    got = gs.labels[:, :n].cpu().numpy()
    for c in range(B):
        agree = (got[c][1:] == ref_labels[c][1:]).mean()
        print(f'clip {c}: label agreement {agree:.5f}, indexes {ge.long_memories_indexes(c)}, drops {ge.drop_trace[c]}')
        assert agree > 0.995
        assert (ge.long_memories_indexes(c), ge.drop_trace[c]) == ref_traces[c]


def _group_vs_fixture(name, B, lookahead):
    """GroupEngine + GroupSlot (hipGraphs, look-ahead encoder on the side stream, label-only post-processing) with B copies of a
    golden clip, the reference's labels fed back (GroupSlot.step(feed=...)): per clip (labels, 1/4-resolution logits resized like
    the reference does at the fixture's sample points [n - 1, 11, points], bank index trace per frame) + the fixture."""
    fitted = 'fitted' in name
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.weights import fitted_state_dict, synth_state_dict
    g, frames, mask, (former, latter, n, h, w, oh, ow, gap, objs) = _load(name)
    dev = torch.device('cuda', 0)
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = former, latter
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(fitted_state_dict(0) if fitted else synth_state_dict(0))
    ge = GroupEngine(model, B, 0, gap, lookahead=lookahead)
    assert ge.use_graphs
    gs = GroupSlot(ge, (oh, ow), dev)
    fd = frames.to(dev)
    gs.start([fd] * B, [mask.to(dev)] * B, int(mask.max()))
    ge.long_term_mem_gap = gap              # (GroupSlot.start sets the evaluator's gap for the clip length; the fixture names its own)
    ys, xs = torch.from_numpy(g['sample_y']).to(dev), torch.from_numpy(g['sample_x']).to(dev)
    gold = torch.from_numpy(g['labels']).to(dev)[:, None].expand(-1, B, -1, -1).contiguous()     # [n - 1, B, oh, ow]
    torch.cuda.synchronize()                # (the engine reads `feed` on its own stream)
    samples, traces = [], [[] for _ in range(B)]
    rt = ge.rt
    while not gs.done:
        i = gs.cursor
        gs.step(feed=gold[i - 1])
        ge.synchronize()
        lg = rt.logits.view(B, rt.H4, rt.W4, 16)[..., :11].permute(0, 3, 1, 2)
        up = F.interpolate(lg, size=(oh, ow), mode='bilinear', align_corners=True)       # aot_engine.py:455-458
        samples.append(up[:, :, ys, xs].cpu().numpy())
        for c in range(B):
            traces[c].append(list(ge.long_memories_indexes(c)))
    return g, gs.labels[:, 1:n].cpu().numpy(), np.stack(samples, 1), traces, ge


@pytest.mark.parametrize('name', ['clip_full.npz', 'clip_full_fitted.npz'])
def test_group_engine_bench_path_vs_reference_fixture(name):
    """The path bench.py times -- GroupEngine + GroupSlot, 8 clips per group, hipGraphs, encoder look-ahead 2 on the side stream,
    label-only post-processing, the LSTT chain kernels -- at BENCH GEOMETRY (480x854 -> 481x849, bank N = 8) against the
    reference's own golden clip (managers/evaluator.py:385-441, 509-523; engines/aot_engine.py:438-465).  All eight clips of the
    group are the fixture's clip; the reference's labels are fed back, so every frame is an independent comparison: per-frame
    labels, mask IoU, logits and the bank index trace must match the fixture with the per-clip tests' tolerances, for every clip."""
    if not os.path.exists(os.path.join(GOLDEN, name)):
        pytest.skip(f'{name} not generated')
    fitted = 'fitted' in name
    if fitted and not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('fitted weights missing')
    B = 8
    g, got, samples, traces, ge = _group_vs_fixture(name, B, 2)
    assert tuple(g['meta'][3:7]) == (481, 849, 480, 854) and int(g['meta'][0] + g['meta'][1]) == 8
    assert ge.rt.chain and ge.rt.pair_attn
    ref = g['logit_samples']
    for c in range(B):
        assert (_trace_matrix(traces[c], g['indexes']) == g['indexes']).all(), c
        err = np.abs(samples[c] - ref).max()
        agree = (got[c] == g['labels']).mean()
        ious = [_iou(a, b) for a, b in zip(g['labels'], got[c])]
        print(f'{name} group clip {c}: max |dlogit| {err:.4f} at logit std {ref.std():.2f}, label agreement {agree:.5f}, mean IoU {np.mean(ious):.5f}')
        if fitted:
            assert err < 0.035 * ref.std() and np.mean(ious) >= 0.99 and agree >= 0.999
        else:
            assert err < 0.065 * ref.std() and agree > 0.97
    # the eight clips ran the same inputs through one launch: identical labels
    assert all(np.array_equal(got[0], got[c]) for c in range(1, B))


@pytest.mark.parametrize('name', ['clip_long_n8.npz', 'clip_long_n2_fitted.npz'])
def test_long_clip_eviction_traces(name):
    """160-frame clips, gap 2: 72 (bank N = 8) / 78 (N = 2) evictions -- the eviction policy's EMA scores and UCB visit counts
    (layers/transformer.py:357-411) far past the ~20 evictions of the short clips (SURVEY.md §8c asks for >= 120 frames).
    Teacher-forced against the reference's clip, on the per-clip engine (hipGraphs) AND on a group of three (throughput path):
    identical bank index trace at every frame, logits within the 16-bit tolerance."""
    if 'fitted' in name and not os.path.exists(os.path.join(GOLDEN, 'trained_delta.pt')):
        pytest.skip('fitted weights missing')
    tol = 0.035 if 'fitted' in name else 0.065
    g, labels, samples, trace = _run(name, True, use_graphs=True)
    ref = g['logit_samples']
    assert (_trace_matrix(trace, g['indexes']) == g['indexes']).all()
    evictions = sum(1 for a, b in zip(trace, trace[1:]) if len(a) == len(b) and a != b)
    err = np.abs(samples - ref).max()
    print(f'{name} per-clip engine: {evictions} evictions, max |dlogit| {err:.4f} at logit std {ref.std():.2f}, label agreement {(labels == g["labels"]).mean():.5f}')
    assert evictions >= 60 and err < tol * ref.std()
    g, got, gsamples, traces, ge = _group_vs_fixture(name, 3, 2)
    for c in range(3):
        assert (_trace_matrix(traces[c], g['indexes']) == g['indexes']).all(), c
        assert np.abs(gsamples[c] - ref).max() < tol * ref.std()
        assert len(ge.drop_trace[c]) == evictions


def _per_clip_reference(former, latter, gap, frames, mask, objs, out_hw, new_object=None, model_name='r50_aotl'):
    """One clip through the drop-in per-clip engine with the evaluator's protocol (propagate -> argmax -> update, or re-add the
    frame as a reference frame when a new object's mask arrives, managers/evaluator.py:484-508): labels and the bank trace."""
    dev = torch.device('cuda', 0)
    eng = _engine(former, latter, gap, model_name=model_name)
    fd = frames.to(dev)
    eng.add_reference_frame(fd[0:1], mask.to(dev), obj_nums=[objs], frame_step=0)
    labels, trace = [], []
    for i in range(1, frames.shape[0]):
        logit = eng.match_propogate_one_frame(fd[i:i + 1], output_size=out_hw)
        label = torch.argmax(logit, dim=1, keepdim=True).float()
        if new_object is not None and new_object[0] == i:
            new = new_object[1].to(dev).float()[None, None]
            label = torch.where(new > 0, new, label)
            eng.add_reference_frame(fd[i:i + 1], F.interpolate(label, size=eng.input_size_2d, mode='nearest'),
                                    obj_nums=[int(label.max().item())], frame_step=i)
        else:
            eng.update_memory(F.interpolate(label, size=eng.input_size_2d, mode='nearest'))
        labels.append(label[0, 0].to(torch.uint8).cpu().numpy())
        trace.append(list(eng.long_memories_indexes))
    return np.stack(labels), trace


def test_group_engine_new_object_in_one_clip():
    """cfg-3 protocol in the throughput mode: three clips in lockstep, a new object's mask arrives at frame 15 of clip 1 only.  That
    clip's bank restarts at one entry and its long-term schedule restarts there, so the group then holds banks of different
    lengths (padded key-table rows) that append at different frames; every clip must deliver what its per-clip engine delivers."""
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    B, n, oh, ow, objs = 3, 30, 160, 192, 2
    clips = [make_clip(80 + c, n, 161, 193, objs) for c in range(B)]
    new = torch.zeros(oh, ow, dtype=torch.uint8)
    new[oh // 2:oh // 2 + oh // 4, ow // 8:ow // 8 + ow // 5] = objs + 1
    refs = [_per_clip_reference(1, 7, 2, f, m, objs, (oh, ow), new_object=(15, new) if c == 1 else None) for c, (f, m) in enumerate(clips)]
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 7
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    ge = GroupEngine(model, B, 0, 2, lookahead=2)
    gs = GroupSlot(ge, (oh, ow), dev)
    gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], objs, new_objects={1: (15, new.to(dev))})
    ge.long_term_mem_gap = 2                       # (GroupSlot.start sets the evaluator's gap for the clip length; the fixture protocol uses 2)
    traces, banks = [[] for _ in range(B)], []
    while not gs.done:
        gs.step()
        for c in range(B):
            traces[c].append(list(ge.long_memories_indexes(c)))
        banks.append([len(sl) for sl in ge.rt.slots])
    ge.synchronize()
    got = gs.labels[:, :n].cpu().numpy()
    assert banks[14] == [8, 1, 8] and banks[16] == [8, 2, 8], banks     # after frame 15 clip 1 holds one entry and appends on its own schedule
    for c in range(B):
        agree = (got[c][1:] == refs[c][0]).mean()
        print(f'clip {c}: label agreement {agree:.5f}, final indexes {traces[c][-1]}')
        assert agree > 0.995
        assert traces[c] == refs[c][1], (c, traces[c][-1], refs[c][1][-1])
    assert (got[1][15] == objs + 1).sum() > 0


def test_group_engine_unbounded_bank():
    """cfg-4 protocol in the throughput mode: latter_mem_len = 9999 (tools/eval.py:92), the banks of both clips grow to 20 entries
    and nothing is ever evicted; masks and bank traces of the per-clip engines."""
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    B, n, oh, ow, objs = 2, 40, 160, 192, 2
    clips = [make_clip(90 + c, n, 161, 193, objs) for c in range(B)]
    refs = [_per_clip_reference(1, 9999, 2, f, m, objs, (oh, ow)) for f, m in clips]
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 9999
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    ge = GroupEngine(model, B, 0, 2, lookahead=2)
    gs = GroupSlot(ge, (oh, ow), dev)
    gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], objs)
    ge.long_term_mem_gap = 2
    while not gs.done:
        gs.step()
    ge.synchronize()
    got = gs.labels[:, :n].cpu().numpy()
    for c in range(B):
        agree = (got[c][1:] == refs[c][0]).mean()
        print(f'clip {c}: label agreement {agree:.5f}, bank {len(ge.long_memories_indexes(c))} entries')
        assert agree > 0.995
        assert ge.long_memories_indexes(c) == refs[c][1][-1] and len(refs[c][1][-1]) == 20


def test_group_engine_table_uploads_never_race_the_gpu():
    """The key table and the append-slot table are re-sent through small pinned staging rings while earlier frames are still
    queued (graph replay enqueues ~10x faster than the GPU runs).  Unbounded bank + gap 1: both tables change EVERY frame and
    nothing makes the host wait (no eviction read-back), so a staging row would be rewritten under a queued copy if the ring did
    not wait on its upload events.  A run that never synchronises must equal a run that synchronises after every step."""
    from rmem_ocu_amd import build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    B, n, oh, ow = 2, 30, 160, 192
    clips = [make_clip(120 + c, n, 161, 193, 2) for c in range(B)]
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 9999
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    runs = []
    for sync_every_step in (True, False):
        ge = GroupEngine(model, B, 0, 1, lookahead=2)
        gs = GroupSlot(ge, (oh, ow), dev)
        gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], 2)
        ge.long_term_mem_gap = 1
        if not sync_every_step:              # first build every graph (bank sizes 1..30) so that the racing run only replays
            while not gs.done:
                gs.step()
            ge.synchronize()
            gs.start([f.to(dev) for f, _ in clips], [m.to(dev) for _, m in clips], 2)
            ge.long_term_mem_gap = 1
        while not gs.done:
            gs.step()
            if sync_every_step:
                ge.synchronize()
        ge.synchronize()
        runs.append((gs.labels[:, :n].cpu().numpy().copy(), [list(ge.long_memories_indexes(c)) for c in range(B)]))
    assert runs[0][1] == runs[1][1] and len(runs[0][1][0]) == n
    assert np.array_equal(runs[0][0], runs[1][0]), 'asynchronous table uploads changed the masks'


def test_group_slot_from_pinned_uint8_frames():
    """Clip group fed from decoded uint8 frames in pinned host memory (H2D + ingest kernel into the look-ahead encoder's input)
    gives the masks of the same group fed with the ingested fp32 frames from device memory."""
    from rmem_ocu_amd import build_vos_model, get_config, ops
    from rmem_ocu_amd.clip_runner import GroupSlot
    from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
    from rmem_ocu_amd.synth import make_clip
    from rmem_ocu_amd.weights import synth_state_dict
    dev = torch.device('cuda', 0)
    B, n = 2, 9
    cfg = get_config('pre_vost', 'test', 'r50_aotl')
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = 1, 2
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(0)
    model.load_state_dict(synth_state_dict(0))
    u8s, ings, masks = [], [], []
    for c in range(B):
        f, m = make_clip(60 + c, n, 161, 193, 2)
        vid = F.interpolate(f, size=(160, 192), mode='bilinear', align_corners=False)
        u8 = (vid * 40.0 + 128.0).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().pin_memory()
        u8d = u8.to(dev)
        ing = torch.empty(n, 3, 161, 193, dtype=torch.float32, device=dev)
        ops.run([ops.ingest_rgb8(u8d[i], Hs=160, Ws=192, Hd=161, Wd=193, out_chw=ing[i]) for i in range(n)])
        u8s.append(u8); ings.append(ing); masks.append(m.to(dev))
    torch.cuda.synchronize()
    out = []
    for src in (ings, u8s):
        ge = GroupEngine(model, B, 0, 5, lookahead=2)
        gs = GroupSlot(ge, (160, 192), dev)
        gs.start(src, masks, 2)
        while not gs.done:
            gs.step()
        ge.synchronize()
        out.append(gs.labels[:, :n].cpu().numpy().copy())
    assert np.array_equal(out[0][:, 1:], out[1][:, 1:])
