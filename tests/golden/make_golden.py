"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Container-only: needs /root/reference (read-only) and never runs on the GPU box.
    python tests/golden/make_golden.py [ops|small|full|all]

The reference is imported unmodified.  Three accommodations live only here
(SURVEY.md §8c): in-memory ``sys.modules`` entries for ``timm.models.layers`` and
``torchvision.transforms`` (the reference imports an init helper and an enum it never
uses on this path); the config is composed the way configs/pre_vost.py:16 +
tools/eval.py:134-135 would (``pre_vost`` itself cannot import in this fork); and
``torch.zeros(device=cuda)`` is mapped to CPU for aot_engine.py:209-213.

Weights come from rmem_ocu_amd.weights.synth_state_dict (no checkpoint exists
offline); inputs from rmem_ocu_amd.synth.make_clip or seeded numpy streams, so the
fixtures store only outputs plus the seeds/checksums of the inputs.
"""
from __future__ import annotations

import hashlib
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/aot_plus'

from rmem_ocu_amd.synth import make_clip  # noqa: E402
from rmem_ocu_amd.weights import synth_state_dict  # noqa: E402


def load_reference(former=1, latter=7, encoder='resnet50', fitted=False, model_name='r50_aotl'):
    sys.path.insert(0, REF)
    tml = types.ModuleType('timm.models.layers')
    tml.trunc_normal_ = lambda t, mean=0., std=1., a=-2., b=2.: torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)
    tml.DropPath = torch.nn.Identity
    tml.to_2tuple = lambda x: (x, x)
    tm = types.ModuleType('timm.models')
    tm.layers = tml
    timm = types.ModuleType('timm')
    timm.models = tm
    sys.modules.update({'timm': timm, 'timm.models': tm, 'timm.models.layers': tml})
    tvf = types.ModuleType('torchvision.transforms.functional')
    tvt = types.ModuleType('torchvision.transforms')
    tvt.functional = tvf
    tvt.InterpolationMode = type('InterpolationMode', (), {'BILINEAR': 'bilinear', 'NEAREST': 'nearest'})
    tv = types.ModuleType('torchvision')
    tv.transforms = tvt
    sys.modules.update({'torchvision': tv, 'torchvision.transforms': tvt, 'torchvision.transforms.functional': tvf})

    cfg = importlib.import_module('configs.default').EngineConfig('golden', model_name)   # r50_aotl | r50_deaotl
    cfg.MODEL_LINEAR_Q = False
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = former, latter
    if encoder == 'swin_base':
        # cfg 5: configs/models/swinb_aotl.py lacks every RMem attribute AOT.__init__ reads, so the R50 config is kept and
        # only the three encoder-related values of swinb_aotl.py:10-13 are overlaid
        cfg.MODEL_ENCODER, cfg.MODEL_ALIGN_CORNERS, cfg.MODEL_ENCODER_DIM = 'swin_base', False, [128, 256, 512, 512]

    _zeros = torch.zeros

    def zeros_cpu(*a, **k):
        dev = k.get('device')
        if dev is not None and torch.device(dev).type == 'cuda':
            k['device'] = 'cpu'
        return _zeros(*a, **k)
    torch.zeros = zeros_cpu

    from networks.models import build_vos_model
    from networks.engines import build_engine
    model = build_vos_model(cfg.MODEL_VOS, cfg).eval()
    from rmem_ocu_amd.weights import fitted_state_dict
    kind = 'deaot' if model_name == 'r50_deaotl' else 'aot'
    missing = model.load_state_dict(fitted_state_dict(0) if fitted else synth_state_dict(0, encoder=encoder, model=kind), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return cfg, model, build_engine


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def seeded(seed, shape, scale=1.0):
    rng = np.random.Generator(np.random.PCG64([seed, 0xC0FFEE]))
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32) * np.float32(scale))


# --------------------------------------------------------------------------
def gen_ops(model):
    """Op-level fixtures: the reference's own attention / LSTT-block modules on seeded inputs."""
    out = {}
    lstt = model.LSTT
    blk = lstt.layers[0]
    temporal = torch.cat((model.cur_pos_emb, model.mem_pos_emb), dim=0).detach()
    h, w = 6, 7
    L, C = h * w, 256
    pos = model.pos_generator(torch.zeros(1, C, h, w)).view(1, C, L).permute(2, 0, 1).contiguous()
    out['sine_pos_6x7'] = pos[:, 0].numpy()
    big = model.pos_generator(torch.zeros(1, C, 31, 54)).view(1, C, 31 * 54).permute(2, 0, 1)
    out['sine_pos_31x54_rows'] = big[[0, 53, 54, 800, 1673], 0].numpy()
    with torch.no_grad():
        for T in (1, 2, 4, 5, 8, 9, 12):
            q = seeded(100 + T, (L, 1, C))
            k = seeded(200 + T, (T * L, 1, C))
            v = seeded(300 + T, (T * L, 1, C))
            o, attn = blk.long_term_attn(q, k, v, is_return_attn_weight=True)
            out[f'mha_T{T}_out'] = o[:, 0].numpy()
            out[f'mha_T{T}_mass'] = attn.view(1, 8, L, T, L).mean(1)[0].sum(2).numpy()
            # whole block, propagate mode (curr_id_emb None), with the temporal PE
            tgt = seeded(400 + T, (L, 1, C))
            long_mem = [seeded(500 + T, (T, L, 1, C)), seeded(600 + T, (T, L, 1, C))]
            short_mem = [seeded(700 + T, (L, 1, C)), seeded(800 + T, (L, 1, C))]
            y, mems = blk(tgt, long_mem, short_mem, curr_id_emb=None, self_pos=pos, size_2d=(h, w),
                          temporal_encoding=temporal, save_atten_weights=True)
            out[f'blk_T{T}_out'] = y[:, 0].numpy()
            out[f'blk_T{T}_curK'] = mems[0][0][:, 0].numpy()
            out[f'blk_T{T}_locK'] = mems[2][0][:, 0].numpy()
            out[f'blk_T{T}_locV'] = mems[2][1][:, 0].numpy()
            out[f'blk_T{T}_mass'] = blk.record_attn_weight.numpy()
        # reference-frame mode (curr_id_emb given) -> SDPA paths
        tgt = seeded(900, (L, 1, C))
        idemb = seeded(901, (L, 1, C), 0.5)
        y, mems = blk(tgt, None, None, curr_id_emb=idemb, self_pos=pos, size_2d=(h, w), temporal_encoding=temporal)
        out['blk_ref_out'] = y[:, 0].numpy()
        out['blk_ref_gV'] = mems[1][1][0, :, 0].numpy()
        out['blk_ref_locV'] = mems[2][1][:, 0].numpy()
        # cfg-2 sized memory read (HW = 1674, T = 8): sampled rows + the full mass matrix
        L2, T = 1674, 8
        q = seeded(1000, (L2, 1, C))
        k = seeded(1001, (T * L2, 1, C))
        v = seeded(1002, (T * L2, 1, C))
        o, attn = blk.long_term_attn(q, k, v, is_return_attn_weight=True)
        rows = np.arange(0, L2, 27)
        out['mha_big_rows'] = rows
        out['mha_big_out'] = o[rows, 0].numpy()
        out['mha_big_mass'] = attn.view(1, 8, L2, T, L2).mean(1)[0].sum(2).numpy()
        del attn
        # temporal slot table as the reference computes it (transformer.py:606-621)
        for T in range(2, 33):
            pe = torch.arange(4, dtype=torch.float32).view(1, 1, 4)
            if T <= 4:
                s = F.interpolate(pe[:, :, :T], size=T, mode='linear', align_corners=True)
            else:
                s = torch.flip(F.interpolate(torch.flip(F.interpolate(pe, size=4, mode='linear', align_corners=True), dims=(-1,)), size=T, mode='nearest'), dims=(-1,))
            out[f'slots_T{T}'] = s.view(-1).round().to(torch.int64).numpy()
        # encoder / id-bank / decoder at a small size
        img = seeded(1100, (1, 3, 97, 129))
        xs = model.encode_image(img)
        for i, x in enumerate(xs):
            out[f'enc_x{i}'] = x[0, :, ::3, ::3].numpy() if i < 2 else x[0].numpy()
        embs = [seeded(1200 + i, (7 * 9, 1, 256)) for i in range(3)]
        out['dec_logits'] = model.decode_id_logits(embs, xs)[0].numpy()
        mask = torch.zeros(1, 1, 97, 129, dtype=torch.int32)
        mask[:, :, 10:50, 20:70] = 1
        mask[:, :, 40:90, 60:120] = 3
        oh = (mask == torch.arange(11).view(1, -1, 1, 1)).float()
        oh = torch.cat((oh, torch.zeros(1, 1, 97, 129)), 1)
        out['id_emb'] = model.get_id_emb(oh)[0].numpy()
    return out


def new_object_label(out_hw, obj_id):
    """Synthetic 'new object appears' annotation at the OUTPUT size (stand-in for a YouTube-VOS mid-clip label)."""
    lab = torch.zeros(1, 1, out_hw[0], out_hw[1])
    lab[:, :, out_hw[0] // 2:out_hw[0] // 2 + out_hw[0] // 4, out_hw[1] // 8:out_hw[1] // 8 + out_hw[1] // 5] = obj_id
    return lab


def run_ref_clip(cfg, model, build_engine, frames, first_mask, out_hw, gap, sample_px, inject_at=-1):
    """Drives the reference engine exactly as evaluator.py:385-441, 509-523 does."""
    engine = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=0, long_term_mem_gap=gap)
    engine.eval()
    engine.long_term_mem_gap = gap
    labels, idx_trace, logit_samples = [], [], []
    import contextlib
    import io
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        engine.add_reference_frame(frames[0:1], first_mask, frame_step=0, obj_nums=[int(first_mask.max())])
        for i in range(1, frames.shape[0]):
            logit = engine.match_propogate_one_frame(frames[i:i + 1], output_size=out_hw)
            prob = torch.softmax(logit, dim=1)
            label = torch.argmax(prob, dim=1, keepdim=True).float()
            if i == inject_at:      # evaluator.py:484-508: a new object's annotation arrives on this frame
                new = new_object_label(out_hw, int(first_mask.max()) + 1)
                keep = (new == 0).float()
                label = label * keep + new * (1 - keep)
                engine.add_reference_frame(frames[i:i + 1], F.interpolate(label, size=engine.input_size_2d, mode='nearest'),
                                           obj_nums=[int(label.max().item())], frame_step=i)
            else:
                engine.update_memory(F.interpolate(label, size=engine.input_size_2d, mode='nearest'))
            labels.append(label[0, 0].to(torch.uint8).numpy())
            idx_trace.append(list(engine.aot_engines[0].long_memories_indexes))
            logit_samples.append(logit[0][:, sample_px[0], sample_px[1]].numpy().copy())
    return np.stack(labels), idx_trace, np.stack(logit_samples)


def pack_trace(trace, width):
    arr = -np.ones((len(trace), width), dtype=np.int32)
    for i, t in enumerate(trace):
        arr[i, :len(t)] = t
    return arr


def gen_swin_ops():
    """Swin-B encoder (cfg 5) on a 96x128 image: stage outputs of the reference module."""
    _, model, _ = load_reference(encoder='swin_base')
    out = {}
    with torch.no_grad():
        xs = model.encode_image(seeded(2100, (1, 3, 96, 128)))
        for i, x in enumerate(xs):
            out[f'swin_x{i}'] = x[0, :, ::2, ::2].numpy() if i < 2 else x[0].numpy()
    return out


def gen_deaot_ops(model):
    """Op-level fixtures of the DeAOT propagation block: the reference's own GatedPropagation / LocalGatedPropagation /
    GatedPropagationModule modules (layers/attention.py:93-413, layers/transformer.py:1011-1249) on seeded inputs."""
    out = {}
    temporal = torch.cat((model.cur_pos_emb, model.mem_pos_emb), dim=0).detach()
    h, w = 9, 11
    L, C = h * w, 256
    with torch.no_grad():
        for li in (0, 1):
            blk = model.LSTT.layers[li]
            for T in (1, 3, 5, 9):
                tgt = seeded(3000 + 10 * T + li, (L, 1, C))
                tgt_id = None if li == 0 else seeded(3100 + T, (L, 1, C))
                long_mem = [seeded(3200 + T, (T, L, 1, 128)), torch.nn.functional.silu(seeded(3300 + T, (T, L, 1, 512))), None,
                            torch.nn.functional.silu(seeded(3400 + T, (T, L, 1, 512)))]
                short_mem = [seeded(3500 + T, (1, 128, h, w)), torch.nn.functional.silu(seeded(3600 + T, (1, 512, h, w))), None,
                             torch.nn.functional.silu(seeded(3700 + T, (1, 512, h, w)))]
                y, yid, mems = blk(tgt, tgt_id, long_mem, short_mem, curr_id_emb=None, size_2d=(h, w),
                                   temporal_encoding=temporal, save_atten_weights=True)
                out[f'gpm{li}_T{T}_out'] = y[:, 0][::3].numpy()
                out[f'gpm{li}_T{T}_outid'] = yid[:, 0][::3].numpy()
                out[f'gpm{li}_T{T}_curK'] = mems[0][0][:, 0][::3].numpy()
                out[f'gpm{li}_T{T}_curV'] = mems[0][1][:, 0][::3].numpy()
                out[f'gpm{li}_T{T}_mass'] = blk.record_attn_weight.numpy()
            # reference-frame mode (curr_id_emb given) -> SDPA path, memories of the frame itself
            tgt = seeded(3800 + li, (L, 1, C))
            tgt_id = None if li == 0 else seeded(3810, (L, 1, C))
            idemb = seeded(3820, (L, 1, C), 0.5)
            y, yid, mems = blk(tgt, tgt_id, None, None, curr_id_emb=idemb, size_2d=(h, w), temporal_encoding=temporal)
            out[f'gpm{li}_ref_out'] = y[:, 0].numpy()
            out[f'gpm{li}_ref_outid'] = yid[:, 0].numpy()
            out[f'gpm{li}_ref_gIDV'] = mems[1][3][0, :, 0].numpy()
        blk = model.LSTT.layers[0]
        # the two attention modules alone
        for T in (1, 4, 9):
            q = seeded(4000 + T, (L, 1, 128))
            k = seeded(4100 + T, (T * L, 1, 128))
            v = seeded(4200 + T, (T * L, 1, 1024))
            u = seeded(4300 + T, (L, 1, 1024))
            o, attn = blk.long_term_attn(q, k, v, u, (h, w), is_return_attn_weight=True)
            out[f'gp_T{T}_out'] = o[:, 0].numpy()
            out[f'gp_T{T}_mass'] = attn.view(1, 1, L, T, L).mean(1)[0].sum(2).numpy()
        q2 = seeded(4400, (1, 128, h, w))
        k2 = seeded(4401, (1, 128, h, w))
        v2 = seeded(4402, (1, 1024, h, w))
        u2 = seeded(4403, (L, 1, 1024))
        o, _ = blk.short_term_attn(q2, k2, v2, u2, (h, w))
        out['lgp_out'] = o[:, 0].numpy()
        # a map larger than the 15x15 window in both directions (cfg-2 token grid is 31x54)
        hb, wb = 18, 23
        q2 = seeded(4410, (1, 128, hb, wb))
        k2 = seeded(4411, (1, 128, hb, wb))
        v2 = seeded(4412, (1, 1024, hb, wb))
        u2 = seeded(4413, (hb * wb, 1, 1024))
        o, _ = blk.short_term_attn(q2, k2, v2, u2, (hb, wb))
        out['lgp_big_out'] = o[::2, 0].numpy()
        x = seeded(4500, (L, 1, 512))
        o, _ = blk.self_attn(x, x, x, x, (h, w))
        out['gp_self_out'] = o[:, 0].numpy()
        # identity embedding with id_norm, and the decoder on a 512-wide input (models/deaot.py:56-68)
        img = seeded(1100, (1, 3, 97, 129))
        xs = model.encode_image(img)
        embs = [seeded(4600 + i, (7 * 9, 1, 512)) for i in range(3)]
        out['dec_logits'] = model.decode_id_logits(embs, xs)[0].numpy()
        mask = torch.zeros(1, 1, 97, 129, dtype=torch.int32)
        mask[:, :, 10:50, 20:70] = 1
        mask[:, :, 40:90, 60:120] = 3
        oh = (mask == torch.arange(11).view(1, -1, 1, 1)).float()
        oh = torch.cat((oh, torch.zeros(1, 1, 97, 129)), 1)
        out['id_emb'] = model.get_id_emb(oh)[0].numpy()
    return out


def gen_clip(tag, former, latter, n_frames, h, w, out_hw, gap, objs, seed, inject_at=-1, encoder='resnet50', fitted=False,
             model_name='r50_aotl'):
    cfg, model, build_engine = load_reference(former, latter, encoder, fitted, model_name)
    frames, mask = make_clip(seed, n_frames, h, w, objs)
    ys = np.linspace(2, out_hw[0] - 3, 12).astype(np.int64)
    xs = np.linspace(2, out_hw[1] - 3, 12).astype(np.int64)
    labels, trace, samples = run_ref_clip(cfg, model, build_engine, frames, mask, out_hw, gap, (ys, xs), inject_at)
    width = max(len(t) for t in trace)
    return {
        'meta': np.array([former, latter, n_frames, h, w, out_hw[0], out_hw[1], gap, objs, seed], dtype=np.int64),
        'inject_at': np.array(inject_at),
        'frames_sha': np.array(sha(frames)), 'mask_sha': np.array(sha(mask)),
        'labels': labels, 'indexes': pack_trace(trace, width),
        'sample_y': ys, 'sample_x': xs, 'logit_samples': samples,
    }


def gen_iou():
    """J (region similarity) of the reference's own metric, evaluation/source/metrics.py:6-37, on seeded mask pairs.  The module
    imports cv2 at the top (absent here; only db_eval_boundary, the F metric, uses it): an empty in-memory module entry lets the
    import succeed, db_eval_iou itself is plain numpy."""
    import types
    sys.modules.setdefault('cv2', types.ModuleType('cv2'))
    sys.path.insert(0, os.path.join(os.path.dirname(REF), 'evaluation', 'source') if os.path.basename(REF) == 'aot_plus' else REF)
    import importlib
    metrics = importlib.import_module('metrics')
    rng = np.random.Generator(np.random.PCG64([77, 0xC0FFEE]))
    out = {'n': np.array(24)}
    for i in range(24):
        h, w = int(rng.integers(8, 97)), int(rng.integers(8, 129))
        ids = int(rng.integers(1, 6))
        gt = (rng.integers(0, ids + 1, size=(h // 4 + 1, w // 4 + 1)).repeat(4, 0).repeat(4, 1)[:h, :w]).astype(np.uint8)
        flip = rng.random((h, w)) < 0.15
        pred = np.where(flip, rng.integers(0, ids + 1, size=(h, w)), gt).astype(np.uint8)
        if i % 6 == 0:
            pred[pred == 1] = 0; gt[gt == 1] = 0                    # an id absent from both: union 0 -> J = 1
        void = (rng.random((h, w)) < 0.05) if i % 3 == 0 else None
        js = [float(metrics.db_eval_iou(gt == k, pred == k, void)) for k in range(1, ids + 1)]
        out[f'gt{i}'], out[f'pred{i}'], out[f'j{i}'] = gt, pred, np.array(js, dtype=np.float64)
        out[f'void{i}'] = np.zeros((0,), dtype=bool) if void is None else void
    return out


def gen_tta():
    """Test-time-augmentation merge of the reference (managers/evaluator.py:427-441) on seeded logits: per augmentation
    `flip_tensor(pred_logit, 3)` where the engine ran on the flipped frame (utils/image.py:109-113, imported from the reference),
    softmax over the classes, mean over the augmentations, argmax.  Stored: the per-augmentation logits AS THE ENGINES RETURN THEM
    (flipped ones still flipped), the flip flags, the reference's mean probability and label."""
    sys.path.insert(0, REF)
    flip_tensor = importlib.import_module('utils.image').flip_tensor
    rng = np.random.Generator(np.random.PCG64([88, 0xC0FFEE]))
    out = {}
    cases = [([False], 24, 40), ([False, True], 24, 40), ([False, True, False, True], 33, 47), ([True, True, False], 16, 20)]
    out['n'] = np.array(len(cases))
    for i, (flips, h, w) in enumerate(cases):
        lg = torch.from_numpy((rng.standard_normal((len(flips), 11, h, w)) * 2.5).astype(np.float32))
        all_preds = []
        for a, fl in enumerate(flips):
            pred_logit = lg[a:a + 1]
            if fl:
                pred_logit = flip_tensor(pred_logit, 3)
            all_preds.append(torch.softmax(pred_logit, dim=1))
        pred_prob = torch.mean(torch.cat(all_preds, dim=0), dim=0, keepdim=True)
        pred_label = torch.argmax(pred_prob, dim=1, keepdim=True).float()
        out[f'logits{i}'], out[f'flips{i}'] = lg.numpy(), np.array(flips)
        out[f'prob{i}'], out[f'label{i}'] = pred_prob.numpy(), pred_label.numpy().astype(np.uint8)
    return out


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    torch.set_num_threads(8)
    if what in ('tta', 'all'):
        np.savez_compressed(os.path.join(HERE, 'tta.npz'), **gen_tta())
    if what in ('iou', 'all'):
        np.savez_compressed(os.path.join(HERE, 'iou.npz'), **gen_iou())
    if what in ('ops', 'all'):
        _, model, _ = load_reference()
        np.savez_compressed(os.path.join(HERE, 'ops.npz'), **gen_ops(model))
    if what in ('small', 'all'):
        # 161x193 -> 11x13 tokens; bank of 3 with gap 2 so ~20 evictions happen in 48 frames
        np.savez_compressed(os.path.join(HERE, 'clip_small.npz'),
                            **gen_clip('small', 1, 2, 48, 161, 193, (160, 192), 2, 3, 11))
    if what in ('newobj', 'all'):
        # cfg-3 protocol: a new object's mask arrives at frame 15 -> the engine re-adds a reference frame mid-clip
        # (bank re-initialised to one entry while long_memories_indexes keeps growing, aot_engine.py:322-323).
        # N = 8, 30 frames, gap 2: the bank never exceeds 8 entries.  NB with a smaller bank the REFERENCE raises at the
        # first eviction after the injection (layers/transformer.py:401: attn_weight has T' entries, frame_times has
        # len(long_memories_indexes) - 1 > T'), observed here with N = 4, 36 frames.
        np.savez_compressed(os.path.join(HERE, 'clip_newobj.npz'),
                            **gen_clip('newobj', 1, 7, 30, 161, 193, (160, 192), 2, 2, 31, inject_at=15))
    if what in ('unbounded', 'all'):
        # cfg-4 protocol: unbounded memory (latter_mem_len = 9999, tools/eval.py:92): the bank grows to 20 entries
        np.savez_compressed(os.path.join(HERE, 'clip_unbounded.npz'),
                            **gen_clip('unbounded', 1, 9999, 40, 161, 193, (160, 192), 2, 2, 41))
    # NB no fixture for > 10 objects: the reference keeps the clip's LSTT memory inside the shared model
    # (layers/transformer.py:455-463), so its second AOTEngine (objects 11..) overwrites the first one's bank and the run
    # raises at the first eviction (transformer.py:401) -- observed here with 12 objects, 14 frames, bank 1+2.
    if what in ('swin', 'all'):
        np.savez_compressed(os.path.join(HERE, 'swin_ops.npz'), **gen_swin_ops())
        # SwinB-AOTL clip: align_corners False -> network size multiple of 16 (video_transforms.py:616-622), id bank k16 s16
        np.savez_compressed(os.path.join(HERE, 'clip_swin.npz'),
                            **gen_clip('swin', 1, 2, 16, 160, 192, (160, 192), 2, 2, 61, encoder='swin_base'))
    if what in ('fitted', 'all'):
        # the same two geometries with the FITTED weights (tests/golden/train_synth_weights.py): confident masks, so
        # mask IoU is a meaningful parity measure
        np.savez_compressed(os.path.join(HERE, 'clip_small_fitted.npz'),
                            **gen_clip('small_fitted', 1, 2, 48, 161, 193, (160, 192), 2, 3, 11, fitted=True))
        np.savez_compressed(os.path.join(HERE, 'clip_full_fitted.npz'),
                            **gen_clip('full_fitted', 1, 7, 30, 481, 849, (480, 854), 2, 3, 21, fitted=True))
    if what in ('deaot', 'all'):
        # R50-DeAOTL (the model eval_vost.sh:11 runs): block-level fixtures and two clips
        _, model, _ = load_reference(1, 8, model_name='r50_deaotl')
        np.savez_compressed(os.path.join(HERE, 'deaot_ops.npz'), **gen_deaot_ops(model))
        np.savez_compressed(os.path.join(HERE, 'deaot_clip_small.npz'),
                            **gen_clip('deaot_small', 1, 2, 48, 161, 193, (160, 192), 2, 3, 11, model_name='r50_deaotl'))
        # cfg-2 geometry with the shipped bank size 1 + 8 (configs/models/r50_deaotl.py:8-9, eval_vost.sh:28)
        np.savez_compressed(os.path.join(HERE, 'deaot_clip_full.npz'),
                            **gen_clip('deaot_full', 1, 8, 30, 481, 849, (480, 854), 2, 3, 21, model_name='r50_deaotl'))
    if what in ('n2', 'all'):
        # cfg-1 stand-in (BASELINE.json configs[0]: one DAVIS-16 clip, 82 frames, 480p, bank N = 2 = 1 + 1): 82 frames at network size
        # 481x849, one object, gap = max(round(82 / 30), 5) = 5 as the evaluator sets it (evaluator.py:330-335), fitted weights:
        # the bank overflows at every append from frame 10 on, 15 evictions of the only evictable entry
        np.savez_compressed(os.path.join(HERE, 'clip_n2_fitted.npz'),
                            **gen_clip('n2', 1, 1, 82, 481, 849, (480, 854), 5, 1, 71, fitted=True))
    if what in ('long', 'all'):
        # long clips (eviction traces over >= 120 frames): 160 frames at 161x193, gap 2 -> an append every second
        # frame; bank N = 8 (1 + 7): 72 evictions, bank N = 2 (1 + 1): 78 -- the EMA scores and the UCB visit counts of the policy
        # (layers/transformer.py:357-411) run far past the ~20 evictions of the other clips
        np.savez_compressed(os.path.join(HERE, 'clip_long_n8.npz'),
                            **gen_clip('long_n8', 1, 7, 160, 161, 193, (160, 192), 2, 3, 91))
        np.savez_compressed(os.path.join(HERE, 'clip_long_n2_fitted.npz'),
                            **gen_clip('long_n2', 1, 1, 160, 161, 193, (160, 192), 2, 2, 92, fitted=True))
    if what in ('full', 'all'):
        # cfg-2 geometry: 480x854 video at network size 481x849, bank N = 8, gap 2 so the bank fills
        # by frame 14 and evicts from frame 16
        np.savez_compressed(os.path.join(HERE, 'clip_full.npz'),
                            **gen_clip('full', 1, 7, 30, 481, 849, (480, 854), 2, 3, 21))
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')
