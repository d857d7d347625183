"""Per-clip inference engines with the reference's method surface.

``AOTEngine`` mirrors networks/engines/aot_engine.py:18-568 (inference half) and
``AOTInferEngine`` mirrors 571-725: same method names, arguments, attributes
(``long_term_mem_gap, input_size_2d, enc_size_2d, enc_hw, long_memories_indexes``) and
error behaviour; the math runs as HIP launch lists (rmem_ocu_amd.runtime).  What stays on
the host is exactly what the reference keeps in Python: the frame counter, the
"append every ``gap`` frames" rule (338-343), and the restricted-memory eviction policy
(layers/transformer.py:324-436), restated in ``MemoryPolicy``.

Differences by design: clip state is engine-owned (the reference stores it in
the shared model); bs = 1 per engine (the reference asserts it in eval,
layers/transformer.py:641); the attention-weight top-32 D2H of every layer
(transformer.py:644-648) is not produced (only dead code consumed it).
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from ... import ops
from ...runtime import ClipRuntime

F32 = torch.float32


class MemoryPolicy:
    """Which bank entry to evict (layers/transformer.py:338-411, eval branch).

    Inputs are the per-memory-frame attention mass of layer 0 weighted by the
    foreground probability (already reduced over tokens on the device); the EMA (0.8) with the
    stored score of the same frame index, the UCB bonus 1.5*sqrt(log(sum n)/(n_i+8)) with
    n_0 := T', and the argmin over entries >= 1 are evaluated here in fp32 torch CPU ops,
    the same arithmetic the reference runs.
    """

    def __init__(self):
        self.ema: Dict[int, torch.Tensor] = {}
        self.visits: Dict[int, int] = {}

    def choose(self, scores: torch.Tensor, indexes: List[int]) -> int:
        a = (scores / scores.sum()).clone()
        cur = {indexes[i]: a[i].clone() for i in range(a.shape[0])}
        cur = {k: ((1 - 0.8) * self.ema[k] + 0.8 * v) if k in self.ema else v for k, v in cur.items()}
        self.ema = cur
        for i in range(a.shape[0]):
            a[i] = cur[indexes[i]]
        self.visits = {k: 1 + self.visits.get(k, 0) for k in indexes}
        n = torch.tensor([float(self.visits[k]) for k in indexes[:-1]])
        n[0] = len(n)
        a = a + 1.5 * torch.sqrt(torch.log(n.sum()) / (n + 8))
        rest = a[1:]
        return int(torch.argmin(rest).item()) + 1 if rest.shape[0] > 0 else 1


class AOTEngine:
    policy_every_update = False      # DeAOT: scores / visit counts move on every long-term update (deaot_engine.py)

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1):
        self.cfg = aot_model.cfg
        self.align_corners = aot_model.cfg.MODEL_ALIGN_CORNERS
        self.AOT = aot_model
        self.max_obj_num = aot_model.max_obj_num
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.short_term_mem_skip = short_term_mem_skip
        if short_term_mem_skip != 1:
            raise NotImplementedError('short_term_mem_skip != 1 is not used by the reference evaluator')
        self.device = torch.device('cuda', gpu_id)
        # every engine enqueues on its own HIP stream (clips overlap on the GPU; the null stream cannot be
        # captured into a hipGraph); sync_caller orders it after/before the caller's current stream like
        # the reference's default-stream semantics
        self.stream = torch.cuda.Stream(self.device)
        self.sync_caller = True
        self.rt: Optional[ClipRuntime] = None
        self.use_graphs = False
        self._graphs: Dict[str, ops.Graph] = {}
        self.restart_engine()

    # ------------------------------------------------------------------ state
    def restart_engine(self, batch_size=1, enable_id_shuffle=False):
        if batch_size != 1 or enable_id_shuffle:
            raise NotImplementedError('inference engine: batch_size 1, no id shuffle (layers/transformer.py:641)')
        self.batch_size = 1
        self.frame_step = 0
        self.last_mem_step = -1
        self.obj_nums = None
        self.input_size_2d = None
        self.enc_size_2d = None
        self.enc_hw = None
        self._indexes: List[int] = []
        self._pending_evict = None
        self.policy = MemoryPolicy()
        self.drop_trace: List[int] = []
        self.pred_id_logits = None
        self._T_at_propagate = 0
        if self.rt is not None:
            self.rt.reset_bank()

    def eval(self):
        return self

    @property
    def long_memories_indexes(self) -> List[int]:
        """Frame indexes of the bank entries (aot_engine.py:323, 351); resolves a deferred eviction first."""
        self._resolve_pending()
        return self._indexes

    def update_size(self, input_size, enc_size):
        self.input_size_2d = tuple(int(v) for v in input_size)
        self.enc_size_2d = tuple(int(v) for v in enc_size)
        self.enc_hw = self.enc_size_2d[0] * self.enc_size_2d[1]

    def _stream(self) -> int:
        return self.stream.cuda_stream

    @contextlib.contextmanager
    def _scope(self):
        cur = torch.cuda.current_stream(self.device)
        if self.sync_caller:
            self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            yield cur
        if self.sync_caller:
            cur.wait_stream(self.stream)

    def _ensure_runtime(self, img):
        H, W = int(img.shape[-2]), int(img.shape[-1])
        n = self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN
        slots = n + 1 if n < 64 else 16            # +1: the bank holds N+1 entries between append and eviction
        if self.rt is None or (self.rt.H, self.rt.W) != (H, W):
            P = self.AOT.packed()
            if 'g0.qvu.w' in P:
                from ...runtime_deaot import DeAOTRuntime as Runtime
            else:
                Runtime = ClipRuntime
            self.rt = Runtime(P, (H, W), slots, self.device, self.cfg.MODEL_LSTT_NUM, self.align_corners, self.max_obj_num + 1)
            self.img_in = torch.empty(3, H, W, dtype=F32, device=self.device)
            self.label_in = torch.empty(H, W, dtype=F32, device=self.device)
            self._graphs = {}
        return self.rt

    def _run(self, key: str, prog: list):
        """Enqueue a launch list on the engine's stream: directly, or (use_graphs) as ONE hipGraph per key."""
        s = self._stream()
        ab = os.environ.get('RMEM_ABLATE')           # timing experiments only: drop every launch whose name contains the tag
        if ab:
            prog = [o for o in prog if not any(t in o.name + ':' + getattr(o, 'tag', '') for t in ab.split(','))]
        if self.use_graphs:
            key = f'{key}@{self.rt.bank_generation}'
            g = self._graphs.get(key)
            if g is None:
                ops.run(prog, s)               # warm run (first-touch, lazy module load) outside capture
                g = self._graphs[key] = ops.Graph(prog, s)
            else:
                g(s)
        else:
            ops.run(prog, s)

    def _set_label(self, mask):
        m = mask.reshape(mask.shape[-2], mask.shape[-1])
        if tuple(m.shape) != (self.rt.H, self.rt.W):
            raise ValueError(f'mask must be at the network size {(self.rt.H, self.rt.W)}, got {tuple(m.shape)}')
        self._copy_in(self.label_in, m)

    def _copy_in(self, dst, src):
        """Frame / label into the engine's fixed input buffer on the engine's stream: a plain memcpy node through the C ABI when
        the caller's tensor already has the buffer's type and layout (the evaluator's do), torch's converting copy otherwise."""
        if src.is_cuda and src.dtype == dst.dtype and src.is_contiguous():
            ops.copy_async(dst, src, dst.numel() * dst.element_size())(self._stream())
        else:
            dst.copy_(src, non_blocking=True)

    # ------------------------------------------------------------------ reference frame
    def add_reference_frame(self, img=None, mask=None, frame_step=-1, obj_nums=None, img_embs=None):
        if self.obj_nums is None and obj_nums is None:
            print('No objects for reference frame!')
            exit()
        elif obj_nums is not None:
            self.obj_nums = obj_nums
        if frame_step == -1:
            frame_step = self.frame_step
        if img is None:
            print('No image for reference frame!')
            exit()
        if mask is None:
            print('No mask for reference frame!')
            exit()
        # a deferred eviction belongs to the bank that is about to be reset, but its effect on long_memories_indexes
        # (the reference keeps that list across the reset, aot_engine.py:323) and on the policy state must still happen --
        # against the runtime it was issued on: a frame of another size replaces self.rt below
        self._resolve_pending()
        rt = self._ensure_runtime(img)
        if self.input_size_2d is None:
            self.update_size(img.shape[2:], (rt.H16, rt.W16))
        with self._scope():
            self._copy_in(self.img_in, img.reshape(3, rt.H, rt.W))
            self._set_label(mask)
            # (re)initialise the bank to this frame only (aot_engine.py:322; quirk: long_memories_indexes keeps growing, 323)
            rt.prepare_pos(self._stream())
            rt.reset_bank()
            slot = rt.take_slot()
            rt.slots.append(slot)
            rt.upload_chunks(self._stream())
            self._run(f'ref{slot}', rt.prog_encode(self.img_in) + rt.prog_id_emb(self.label_in, rt.H, rt.W) +
                      rt.prog_lstt(True, 1, slot) + rt.prog_decode())
            self.last_mem_step = frame_step
            self.policy = MemoryPolicy()
            self._indexes.append(self.frame_step)
            self.pred_id_logits = rt.logits

    # ------------------------------------------------------------------ propagate
    def match_propogate_one_frame(self, img=None, img_embs=None, mask=None, output_size=None):
        self.frame_step += 1
        if img is None:
            raise ValueError('match_propogate_one_frame needs the frame (offline encoding is a training-only path)')
        rt = self.rt
        self._resolve_pending()
        with self._scope() as cur:
            self._copy_in(self.img_in, img.reshape(3, rt.H, rt.W))
            T = len(rt.slots)
            self._T_at_propagate = T
            wm = self._mass_needed(T)
            self._run(f'prop{T}{int(wm)}', rt.prog_encode(self.img_in) + rt.prog_lstt(False, T, want_mass=wm) + rt.prog_decode())
            self.pred_id_logits = rt.logits
            out = self._logits_out(output_size)
            out.record_stream(cur)
        return out

    def encode_ahead(self, imgs, frames: int = 4):
        """Run the ResNet-50 encoder for the next ``imgs.shape[0] <= frames`` frames of the clip as one launch per layer
        (frames do not depend on each other before the LSTT; the reference's loader has them ready,
        dataloaders/eval_datasets.py:57-64).  propagate_to_label(..., enc_slot=e) then consumes frame e."""
        rt = self.rt
        be = rt.batch_encoder(frames)
        with self._scope():
            if imgs is not None:                      # None: the caller already put the frames into encode_inputs(frames)
                n = int(imgs.shape[0])
                if n > frames:
                    raise ValueError(f'encode_ahead: {n} frames for a look-ahead of {frames}')
                ops.copy_async(be.img_in, imgs, n * 3 * rt.H * rt.W * 4)(self._stream())
            self._run(f'encB{frames}', be.prog())

    def encode_inputs(self, frames: int = 4):
        """fp32 [frames, 3, H, W] input buffer of encode_ahead (e.g. the target of rmem_ingest_rgb8, so decoded uint8 frames
        go host -> device -> resize + normalise -> encoder without an extra copy)."""
        return self.rt.batch_encoder(frames).img_in

    def propagate_to_label(self, img, label_u8, enc_slot=None):
        """Fused fast path of one frame: propagate, then argmax labels (uint8 [Ho, Wo], caller's device
        buffer at a fixed address) straight from the 1/4-resolution logits -- the evaluator's
        softmax -> argmax (managers/evaluator.py:430-441) without materialising [11, Ho, Wo] logits.
        enc_slot: the frame was encoded by encode_ahead (slot index); img is then unused."""
        self.frame_step += 1
        rt = self.rt
        self._resolve_pending()
        Ho, Wo = int(label_u8.shape[-2]), int(label_u8.shape[-1])
        keep = self.obj_nums[0] if self.obj_nums else self.max_obj_num
        with self._scope():
            T = len(rt.slots)
            self._T_at_propagate = T
            pk = f'post_{label_u8.data_ptr()}_{Ho}_{Wo}'
            if pk not in rt._prog:
                rt._prog[pk] = [ops.logits_post(rt.logits, ldl=16, nc=rt.nc, keep=keep, Hi=rt.H4, Wi=rt.W4, Ho=Ho, Wo=Wo,
                                                align_corners=self.align_corners, label_u8=label_u8)]
            wm = self._mass_needed(T)
            if enc_slot is None:
                ops.copy_async(self.img_in, img, 3 * rt.H * rt.W * 4)(self._stream())
                key = f'propl{T}{int(wm)}_{label_u8.data_ptr()}_{Ho}_{Wo}'
                self._run(key, rt.prog_encode(self.img_in) + rt.prog_lstt(False, T, want_mass=wm) + rt.prog_decode() + rt._prog[pk])
            else:
                key = f'propl{T}{int(wm)}e{enc_slot}_{label_u8.data_ptr()}_{Ho}_{Wo}'
                self._run(key, rt.prog_project(enc_slot) + rt.prog_lstt(False, T, want_mass=wm) + rt.prog_decode(enc_slot) + rt._prog[pk])
            self.pred_id_logits = rt.logits

    def _mass_needed(self, T: int) -> bool:
        """The per-memory-frame attention mass of layer 0 (layers/transformer.py:636-643) is only read by the eviction policy,
        i.e. when the update that follows this propagation appends to the bank (aot_engine.py:338-343) and the bank then
        overflows (or, DeAOT, on every append): both are known now, so the other frames skip the reduction."""
        will_append = (not getattr(self.cfg, 'NO_LONG_MEMORY', False)) and \
            (self.frame_step - self.last_mem_step >= self.long_term_mem_gap)
        need = will_append and (self.policy_every_update or T + 1 > self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN)
        self._mass_valid = need
        return need

    def decode_current_logits(self, output_size=None):
        """Logits with unused ids masked (aot_engine.py:450-453), resized to output_size (457-463)."""
        with self._scope() as cur:
            out = self._logits_out(output_size)
            out.record_stream(cur)
        return out

    def _logits_out(self, output_size=None):
        rt = self.rt
        Ho, Wo = (rt.H4, rt.W4) if output_size is None else (int(output_size[0]), int(output_size[1]))
        out = torch.empty(1, rt.nc, Ho, Wo, dtype=F32, device=self.device)
        keep = self.obj_nums[0] if self.obj_nums else self.max_obj_num
        ops.run(ops.logits_post(rt.logits, ldl=16, nc=rt.nc, keep=keep, Hi=rt.H4, Wi=rt.W4, Ho=Ho, Wo=Wo,
                                align_corners=self.align_corners, out=out), self._stream())
        return out

    def predict_current_mask(self, output_size=None, return_prob=False):
        """aot_engine.py:467-483."""
        if output_size is None:
            output_size = self.input_size_2d
        logits = self.decode_current_logits(output_size)
        pred_mask = torch.argmax(logits, dim=1)
        return (pred_mask, torch.softmax(logits, dim=1)) if return_prob else pred_mask

    # ------------------------------------------------------------------ memory update
    def update_short_term_memory(self, curr_mask, curr_id_emb=None, step=0):
        if curr_id_emb is not None:
            raise NotImplementedError('update_short_term_memory takes the label mask')
        rt = self.rt
        if curr_mask.dim() == 4 and curr_mask.shape[1] != 1:
            raise NotImplementedError('probability masks (>10 objects soft aggregation) are not built yet')
        with self._scope():
            self._set_label(curr_mask)
            self._finish_update('lab', rt.prog_id_emb(self.label_in, rt.H, rt.W))
        if self.sync_caller:
            self._resolve_pending()

    def update_memory_from_label_u8(self, label_u8: torch.Tensor):
        """Fast path: argmax labels at the OUTPUT size (uint8 [Ho, Wo], device); the nearest resize to the
        network size (evaluator.py:518-522) happens inside the one-hot kernel."""
        rt = self.rt
        hs, ws = int(label_u8.shape[-2]), int(label_u8.shape[-1])
        with self._scope():
            self._finish_update(f'u8_{label_u8.data_ptr()}', rt.prog_id_emb(label_u8, hs, ws))

    def _finish_update(self, id_key: str, id_prog: list):
        """Memory update of the frame just propagated (aot_engine.py:327-369).  The launch list (identity embedding +
        short-term update [+ bank append]) is one graph; if the bank now exceeds its size the eviction scores are
        reduced on the device and read back asynchronously -- the decision itself is taken lazily
        (_resolve_pending) right before the bank is used again, so the host never waits here."""
        rt, s = self.rt, self._stream()
        update_long = (not getattr(self.cfg, 'NO_LONG_MEMORY', False)) and \
            (self.frame_step - self.last_mem_step >= self.long_term_mem_gap)
        slot = None
        if update_long:
            self.last_mem_step = self.frame_step
            slot = rt.take_slot()
        self._run(f'upd_{slot}_{id_key}', id_prog + rt.prog_update(slot))
        if not update_long:
            return
        rt.slots.append(slot)
        self._indexes.append(self.frame_step)
        n_keep = self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN
        overflow = len(rt.slots) > n_keep
        if overflow or self.policy_every_update:
            if not getattr(self, '_mass_valid', False):
                raise RuntimeError('long_term_mem_gap / memory length changed between match_propogate_one_frame and update_memory: '
                                   'the attention mass of this frame was not recorded')
            Tp = self._T_at_propagate
            keep = self.obj_nums[0] if self.obj_nums else self.max_obj_num
            ops.run([ops.evict_scores(rt.logits, rt.mass, rt.scores, ldl=16, nc=rt.nc, keep=keep, Hi=rt.H4, Wi=rt.W4,
                                      He=rt.H16, We=rt.W16, T=Tp),
                     ops.copy_async(rt.scores_host, rt.scores, 4 * Tp)], s)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._pending_evict = (Tp, ev, overflow)
        if not overflow:
            rt.upload_chunks(s)

    def _resolve_pending(self):
        """Finish a deferred eviction: wait for the score readback (normally long done), run the policy on the host
        (layers/transformer.py:353-411), drop the entry from the slot table and upload the new chunk table."""
        if self._pending_evict is None:
            return
        Tp, ev, overflow = self._pending_evict
        self._pending_evict = None
        ev.synchronize()                      # the one host wait of the policy (the reference syncs here too, transformer.py:353)
        rt = self.rt
        drop = self.policy.choose(rt.scores_host[:Tp].clone(), self._indexes)
        if not overflow:                      # DeAOT: the scores moved, nothing is dropped yet
            return
        self.drop_trace.append(drop)
        rt.free.append(rt.slots.pop(drop))
        del self._indexes[drop]
        rt.upload_chunks(self._stream())


class AOTInferEngine:
    ENGINE = AOTEngine

    def __init__(self, aot_model, gpu_id=0, long_term_mem_gap=9999, short_term_mem_skip=1, max_aot_obj_num=None):
        self.cfg = aot_model.cfg
        self.AOT = aot_model
        if max_aot_obj_num is None or max_aot_obj_num > aot_model.max_obj_num:
            self.max_aot_obj_num = aot_model.max_obj_num
        else:
            self.max_aot_obj_num = max_aot_obj_num
        self.gpu_id = gpu_id
        self.long_term_mem_gap = long_term_mem_gap
        self.short_term_mem_skip = short_term_mem_skip
        self.use_graphs = False
        self.aot_engines: List[AOTEngine] = []
        self._pool: List[AOTEngine] = []
        self.restart_engine()

    def eval(self):
        return self

    def restart_engine(self):
        for engine in self.aot_engines:
            engine.restart_engine()
        self._pool = self.aot_engines + [e for e in self._pool if e not in self.aot_engines]
        self.aot_engines = []
        self.obj_nums = None

    def separate_mask(self, mask):
        """aot_engine.py:604-628 (label-map branch): engine e keeps ids e*10+1 .. (e+1)*10, renumbered from 1 (rmem_split_label)."""
        if mask is None:
            return [None] * len(self.aot_engines)
        if len(self.aot_engines) == 1:
            return [mask]
        if mask.dim() == 3 or mask.shape[0] == 1:
            import ctypes as C
            from ... import _lib
            m = mask.to(F32).contiguous()
            s = torch.cuda.current_stream(m.device).cuda_stream
            out = []
            for idx in range(len(self.aot_engines)):
                o = torch.empty_like(m)
                _lib.check(_lib.lib().rmem_split_label(m.data_ptr(), idx * self.max_aot_obj_num + 1, (idx + 1) * self.max_aot_obj_num,
                                                       o.data_ptr(), m.numel(), C.c_void_p(s)), 'rmem_split_label')
                out.append(o)
            return out
        raise NotImplementedError('probability masks for >10 objects')

    def soft_logit_aggregation(self, all_logits):
        """aot_engine.py:650-673 as one HIP launch (rmem_soft_logit_aggregate)."""
        if len(all_logits) == 1:
            return all_logits[0]
        import ctypes as C
        from ... import _lib
        n = len(all_logits)
        lg = [t.contiguous() for t in all_logits]
        _, nc, H, W = lg[0].shape
        out = torch.empty(1, 1 + n * self.max_aot_obj_num, H, W, dtype=F32, device=lg[0].device)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in lg])
        _lib.check(_lib.lib().rmem_soft_logit_aggregate(ptrs, n, nc, self.max_aot_obj_num, H, W, out.data_ptr(),
                                                        C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)),
                   'rmem_soft_logit_aggregate')
        return out

    def add_reference_frame(self, img, mask, obj_nums, frame_step=-1):
        if isinstance(obj_nums, list):
            obj_nums = obj_nums[0]
        aot_num = max(np.ceil(obj_nums / self.max_aot_obj_num), 1)
        while aot_num > len(self.aot_engines):
            if self._pool:
                eng = self._pool.pop(0)           # reuse device buffers of an earlier clip
                eng.long_term_mem_gap = self.long_term_mem_gap
            else:
                eng = self.ENGINE(self.AOT, self.gpu_id, self.long_term_mem_gap, self.short_term_mem_skip)
            eng.use_graphs = self.use_graphs
            eng.sync_caller = not getattr(self, '_async', False)
            self.aot_engines.append(eng)
        for eng, m in zip(self.aot_engines, self.separate_mask(mask)):
            eng.add_reference_frame(img, m, obj_nums=[self.max_aot_obj_num], frame_step=frame_step)
        self.update_size()

    def match_propogate_one_frame(self, img=None, mask=None, output_size=None):
        all_logits = [e.match_propogate_one_frame(img, mask=mask, output_size=output_size) for e in self.aot_engines]
        return self.soft_logit_aggregation(all_logits)

    def update_memory(self, curr_mask):
        for eng, m in zip(self.aot_engines, self.separate_mask(curr_mask)):
            eng.update_short_term_memory(m)

    # -- fused fast path (single engine, <= 10 objects): labels in, labels out, everything on the engine's stream
    def propagate_to_label(self, img, label_u8, enc_slot=None):
        if len(self.aot_engines) != 1:
            raise NotImplementedError('the fused label path covers clips with <= 10 objects')
        self.aot_engines[0].propagate_to_label(img, label_u8, enc_slot)

    def encode_ahead(self, imgs, frames: int = 4):
        if len(self.aot_engines) != 1:
            raise NotImplementedError('encoder look-ahead covers clips with <= 10 objects')
        self.aot_engines[0].encode_ahead(imgs, frames)

    def encode_inputs(self, frames: int = 4):
        return self.aot_engines[0].encode_inputs(frames)

    def update_memory_from_label_u8(self, label_u8):
        self.aot_engines[0].update_memory_from_label_u8(label_u8)

    def set_async(self, use_graphs=True):
        """Detach the engines from the caller's stream (each clip runs on its own stream) and replay frames as hipGraphs."""
        self.use_graphs = use_graphs
        for e in self.aot_engines + self._pool:
            e.use_graphs = use_graphs
            e.sync_caller = False
        self._async = True

    def synchronize(self):
        for e in self.aot_engines:
            e.stream.synchronize()

    def update_size(self):
        self.input_size_2d = self.aot_engines[0].input_size_2d
        self.enc_size_2d = self.aot_engines[0].enc_size_2d
        self.enc_hw = self.aot_engines[0].enc_hw

    @property
    def long_memories_indexes(self):
        return self.aot_engines[0].long_memories_indexes
