"""Per-sequence evaluation protocol on top of the engines (the caller of the hot path).

Mirrors what networks/managers/evaluator.py:330-568 does for one sequence, without its dataset plumbing:
  * per-sequence gap = max(round(n / 30), 5) (330-335), assigned to every engine (356);
  * one engine per test-time augmentation (342-355): horizontal flip and/or extra scales; every augmentation's
    logits are resized to the original size, flipped back, soft-maxed and averaged, then arg-maxed (427-441)
    -- rmem_tta_merge;
  * a ground-truth label that arrives on a later frame (a new object) is merged over the prediction and the frame is
    re-added as a reference frame to every engine (484-508); otherwise the prediction updates the memory (509-523);
  * masks can be written as palette PNGs (utils/image.py:90-106) and scored with the region similarity J
    (evaluation/source/metrics.py:6-37) -- rmem_mask_iou_counts.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ops
from .networks.engines import build_engine


def _davis_palette() -> List[int]:
    """The 256-colour DAVIS palette (bit-reversal colour map; the same table utils/image.py:8-62 hard-codes)."""
    pal = []
    for i in range(256):
        r = g = b = 0
        c = i
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal += [r, g, b]
    return pal


def save_mask(mask_u8: np.ndarray, path: str, squeeze_idx: Optional[Sequence[int]] = None):
    """utils/image.py:90-101: optional un-squeeze of object ids, then an indexed PNG with the DAVIS palette."""
    from PIL import Image
    mask = np.asarray(mask_u8, dtype=np.uint8)
    if squeeze_idx is not None:
        out = np.zeros_like(mask)
        for idx in range(1, len(squeeze_idx)):
            out += ((mask == idx) * squeeze_idx[idx]).astype(np.uint8)
        mask = out
    im = Image.fromarray(mask).convert('P')
    im.putpalette(_davis_palette())
    im.save(path)


def tta_merge(logits: Sequence[torch.Tensor], flips: Sequence[bool], want_prob: bool = False):
    """logits: per-augmentation [1, nc, H, W] fp32 device tensors -> (label uint8 [H, W], label fp32 [1,1,H,W], prob or None)."""
    n = len(logits)
    nc, H, W = logits[0].shape[1:]
    dev = logits[0].device
    label = torch.empty(H, W, dtype=torch.uint8, device=dev)
    label_f = torch.empty(1, 1, H, W, dtype=torch.float32, device=dev)
    prob = torch.empty(1, nc, H, W, dtype=torch.float32, device=dev) if want_prob else None
    ptrs = (C.c_void_p * n)(*[t.contiguous().data_ptr() for t in logits])
    fl = (C.c_int * n)(*[int(f) for f in flips])
    _lib.check(_lib.lib().rmem_tta_merge(ptrs, fl, n, nc, H, W, label.data_ptr(), label_f.data_ptr(),
                                         None if prob is None else prob.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
               'rmem_tta_merge')
    return label, label_f, prob


def region_similarity(pred_u8: torch.Tensor, gt_u8: torch.Tensor, num_ids: int = 11, void_label: int = 255) -> Dict[int, float]:
    """J per object id for one mask pair (device uint8 tensors of equal shape); ids absent from both masks are skipped."""
    assert pred_u8.dtype == torch.uint8 and gt_u8.dtype == torch.uint8 and pred_u8.shape == gt_u8.shape
    counts = torch.zeros(2 * num_ids, dtype=torch.int64, device=pred_u8.device)
    _lib.check(_lib.lib().rmem_mask_iou_counts(pred_u8.contiguous().data_ptr(), gt_u8.contiguous().data_ptr(), pred_u8.numel(),
                                               num_ids, void_label, counts.data_ptr(),
                                               torch.cuda.current_stream(pred_u8.device).cuda_stream), 'rmem_mask_iou_counts')
    c = counts.cpu().view(num_ids, 2)
    return {i: (1.0 if c[i, 1] == 0 else float(c[i, 0]) / float(c[i, 1])) for i in range(1, num_ids) if c[i, 1] > 0}


class SequenceEvaluator:
    """Runs one sequence through the engine(s) exactly as the reference evaluator would."""

    def __init__(self, model, gpu_id: int = 0, flip: bool = False):
        self.model, self.gpu_id, self.flip = model, gpu_id, flip
        self.cfg = model.cfg
        self.engines = []

    def _engine(self, i):
        while len(self.engines) <= i:
            e = build_engine(self.cfg.MODEL_ENGINE, phase='eval', aot_model=self.model, gpu_id=self.gpu_id,
                             long_term_mem_gap=self.cfg.TEST_LONG_TERM_MEM_GAP)
            self.engines.append(e.eval())
        return self.engines[i]

    def run(self, frames, labels: Dict[int, torch.Tensor], out_hw: Tuple[int, int]) -> List[torch.Tensor]:
        """frames: [n, 3, H, W] fp32 device (network size, normalised) or, for multi-scale testing, a list of such tensors, one
        per entry of TEST_MULTISCALE (each at its own stride-aligned size, synth.network_size(..., scale=s));
        labels: {frame index: [1,1,Ho,Wo] fp32 label map at the ORIGINAL size}; labels[0] is the first-frame annotation,
        later entries are newly appearing objects.  One engine per (scale, flip) pair (evaluator.py:342-355); their logits
        are resized to the original size, un-flipped, soft-maxed and averaged (427-438).
        Returns the uint8 label map of every frame after the first, at the original size."""
        per_scale = list(frames) if isinstance(frames, (list, tuple)) else [frames]
        n = per_scale[0].shape[0]
        gap = max(int(round(n / 30)), 5)
        flips = [False, True] if self.flip else [False]
        augs = [(si, fl) for si in range(len(per_scale)) for fl in flips]      # evaluator.py:342-355 order: scale outer, flip inner
        if len(augs) > 8:
            raise ValueError('at most 8 augmentations (scales x flips)')
        aflips = [fl for _, fl in augs]
        outs: List[torch.Tensor] = []

        def resized(src, size, fl):
            """(mirror along W for a flipped augmentation, then) nearest resize to `size` on the device: rmem_resize_nearest_flip_f32"""
            dst = torch.empty(*src.shape[:-2], int(size[0]), int(size[1]), dtype=torch.float32, device=src.device)
            ops.run(ops.resize_nearest_flip(src.contiguous(), dst, flip=fl))
            return dst

        def frame(a, t):
            si, fl = augs[a]
            img = per_scale[si][t:t + 1]
            return resized(img, img.shape[-2:], True) if fl else img

        for a, (si, fl) in enumerate(augs):
            e = self._engine(a)
            e.restart_engine()
            e.long_term_mem_gap = gap
            # the first frame's label is resized THEN mirrored (the data pipeline flips the resized sample,
            # dataloaders/video_transforms.py MultiRestrictSize), later labels are mirrored then resized (evaluator.py:490-522)
            lab = resized(labels[0], per_scale[si].shape[2:], False)
            if fl:
                lab = resized(lab, lab.shape[-2:], True)
            e.add_reference_frame(frame(a, 0), lab, obj_nums=[int(labels[0].max().item())], frame_step=0)
        for t in range(1, n):
            logits = [self.engines[a].match_propogate_one_frame(frame(a, t), output_size=out_hw) for a in range(len(augs))]
            label_u8, label_f, _ = tta_merge(logits, aflips)
            if t in labels:                                   # evaluator.py:484-508
                new = labels[t]
                keep = (new == 0).float()
                label_f = label_f * keep + new * (1 - keep)
                label_u8 = label_f[0, 0].to(torch.uint8)
                nobj = [int(label_f.max().item())]
                for a, (si, fl) in enumerate(augs):
                    lab = resized(label_f, self.engines[a].input_size_2d, fl)
                    self.engines[a].add_reference_frame(frame(a, t), lab, obj_nums=nobj, frame_step=t)
            else:
                for a, (si, fl) in enumerate(augs):
                    e = self.engines[a]
                    if not fl and len(e.aot_engines) == 1:
                        # the nearest resize to the network size (evaluator.py:518-522) happens inside the one-hot kernel;
                        # a fixed buffer, so the engine's prepared launch list for it is built once
                        if getattr(self, '_lab_u8', None) is None or self._lab_u8.shape != label_u8.shape:
                            self._lab_u8 = torch.empty_like(label_u8)
                        ops.copy_async(self._lab_u8, label_u8, label_u8.numel())(torch.cuda.current_stream(label_u8.device).cuda_stream)
                        e.update_memory_from_label_u8(self._lab_u8)
                    else:
                        e.update_memory(resized(label_f, e.input_size_2d, fl))
            outs.append(label_u8)
        return outs
