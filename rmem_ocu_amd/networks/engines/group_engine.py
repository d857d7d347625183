"""Throughput engine: B clips of equal length advance in lockstep through ONE set of launch lists.

``GroupEngine`` is the batched counterpart of AOTEngine/AOTInferEngine (aot_engine.py) for the case the evaluator spends its
time in: clips with <= 10 objects, label masks fed back, no mid-clip reference frames.  Per clip it keeps exactly the host
state AOTEngine keeps -- bank slot order, ``long_memories_indexes``, the eviction policy's EMA scores and visit counts
(networks/layers/transformer.py:338-411) -- while frame counter, append schedule (aot_engine.py:338-343) and bank size are
shared, because clips of one length get one gap (managers/evaluator.py:330-335).  All device work goes through
rmem_ocu_amd.group_runtime.GroupRuntime: one launch per layer for the whole group.  Everything else (other protocols, DeAOT,
Swin, > 10 objects) uses the per-clip engines, which are the drop-in API.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from ... import ops
from ...group_runtime import GroupRuntime
from .aot_engine import MemoryPolicy

F32 = torch.float32


class GroupEngine:
    def __init__(self, aot_model, clips: int, gpu_id: int = 0, long_term_mem_gap: int = 9999, lookahead: int = 4):
        self.cfg = aot_model.cfg
        self.AOT = aot_model
        self.B = clips
        self.gpu_id = gpu_id
        self.device = torch.device('cuda', gpu_id)
        self.align_corners = self.cfg.MODEL_ALIGN_CORNERS
        self.max_obj_num = aot_model.max_obj_num
        self.long_term_mem_gap = long_term_mem_gap
        self.lookahead = lookahead
        self.stream = torch.cuda.Stream(self.device)
        self.use_graphs = True
        self.rt: Optional[GroupRuntime] = None
        self._graphs: Dict[str, ops.Graph] = {}
        self.restart_engine()

    # ------------------------------------------------------------------ state
    def restart_engine(self):
        self.frame_step = 0
        self.last_mem_step = -1
        self.obj_nums = None
        self._indexes: List[List[int]] = [[] for _ in range(self.B)]
        self.policies = [MemoryPolicy() for _ in range(self.B)]
        self.drop_trace: List[List[int]] = [[] for _ in range(self.B)]
        self._pending = None
        self._mass_valid = False
        self._T_at_propagate = 0
        if self.rt is not None:
            self.rt.reset_bank()

    def long_memories_indexes(self, clip: int) -> List[int]:
        self._resolve_pending()
        return self._indexes[clip]

    def _s(self) -> int:
        return self.stream.cuda_stream

    def _ensure_runtime(self, H: int, W: int):
        n = self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN
        if n >= 64:
            raise NotImplementedError('GroupEngine: restricted memory banks only (unbounded memory runs on the per-clip engine)')
        if self.rt is None or (self.rt.H, self.rt.W) != (H, W):
            self.rt = GroupRuntime(self.AOT.packed(), (H, W), n + 1, self.device, self.B, self.cfg.MODEL_LSTT_NUM, self.align_corners,
                                   self.max_obj_num + 1, self.lookahead)
            self.label_in = torch.empty(self.B, H, W, dtype=F32, device=self.device)
            self._graphs = {}
        return self.rt

    def _run(self, key: str, prog: list):
        s = self._s()
        if self.use_graphs:
            g = self._graphs.get(key)
            if g is None:
                ops.run(prog, s)
                self._graphs[key] = ops.Graph(prog, s)
            else:
                g(s)
        else:
            ops.run(prog, s)

    # ------------------------------------------------------------------ reference frames (aot_engine.py:241-325, all clips at once)
    def add_reference_frames(self, imgs: torch.Tensor, masks: torch.Tensor, obj_nums: int):
        """imgs [B, 3, H, W] fp32, masks [B, 1, H, W] label maps at the network size (device)."""
        B = self.B
        H, W = int(imgs.shape[-2]), int(imgs.shape[-1])
        rt = self._ensure_runtime(H, W)
        self.obj_nums = [self.max_obj_num]            # AOTInferEngine forces this (aot_engine.py:697)
        self._pending = None
        with torch.cuda.stream(self.stream):
            s = self._s()
            ops.copy_async(rt.enc_now.img_in, imgs.contiguous(), B * 3 * H * W * 4)(s)
            self.label_in.copy_(masks.reshape(B, H, W), non_blocking=True)
            rt.prepare_pos(s)
            rt.reset_bank()
            first = []
            for c in range(B):
                sl = rt.free[c].pop(0)
                rt.slots[c].append(sl)
                first.append(sl)
            rt.upload_chunks(s)
            rt.upload_append_slots(first, s)
            self._run('ref', rt.prog_encode() + rt.prog_id_emb(self.label_in, H, W) + rt.prog_project(None) + rt.prog_lstt(True, 1) +
                      rt.prog_decode(None))
        self.last_mem_step = self.frame_step
        self.policies = [MemoryPolicy() for _ in range(B)]
        for c in range(B):
            self._indexes[c].append(self.frame_step)

    # ------------------------------------------------------------------ look-ahead encoder
    def encode_inputs(self) -> torch.Tensor:
        """fp32 [lookahead * B, 3, H, W]: frame e of clip c goes to row e * B + c."""
        return self.rt.enc_ahead.img_in

    def encode_ahead(self):
        with torch.cuda.stream(self.stream):
            self._run('encB', self.rt.enc_ahead.prog())

    # ------------------------------------------------------------------ propagate (aot_engine.py:398-465 + evaluator.py:430-441)
    def _mass_needed(self, T: int) -> bool:
        will_append = self.frame_step - self.last_mem_step >= self.long_term_mem_gap
        need = will_append and T + 1 > self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN
        self._mass_valid = need
        return need

    def propagate_to_labels(self, labels_u8: torch.Tensor, enc_slot: Optional[int] = None, imgs: Optional[torch.Tensor] = None):
        """labels_u8: uint8 [B, Ho, Wo] device buffer at a fixed address.  Either enc_slot (frame encoded by encode_ahead) or
        imgs [B, 3, H, W] (encoded now)."""
        self.frame_step += 1
        rt, B = self.rt, self.B
        self._resolve_pending()
        Ho, Wo = int(labels_u8.shape[-2]), int(labels_u8.shape[-1])
        keep = self.obj_nums[0]
        with torch.cuda.stream(self.stream):
            T = rt.T
            self._T_at_propagate = T
            wm = self._mass_needed(T)
            pk = f'post_{labels_u8.data_ptr()}_{Ho}_{Wo}'
            if pk not in rt._prog:
                rt._prog[pk] = [ops.logits_post(rt.logits, ldl=16, nc=rt.nc, keep=keep, Hi=rt.H4, Wi=rt.W4, Ho=Ho, Wo=Wo,
                                                align_corners=self.align_corners, label_u8=labels_u8, images=B)]
            if enc_slot is None:
                ops.copy_async(rt.enc_now.img_in, imgs.contiguous(), B * 3 * rt.H * rt.W * 4)(self._s())
                prog = rt.prog_encode() + rt.prog_project(None) + rt.prog_lstt(False, T, wm) + rt.prog_decode(None) + rt._prog[pk]
            else:
                prog = rt.prog_project(enc_slot) + rt.prog_lstt(False, T, wm) + rt.prog_decode(enc_slot) + rt._prog[pk]
            self._run(f'prop{T}{int(wm)}e{enc_slot}_{labels_u8.data_ptr()}', prog)

    # ------------------------------------------------------------------ memory update (aot_engine.py:327-369)
    def update_from_labels(self, labels_u8: torch.Tensor):
        """labels_u8: uint8 [B, Ho, Wo] argmax labels at the output size (nearest-resized to the network size on the device)."""
        rt, B = self.rt, self.B
        hs, ws = int(labels_u8.shape[-2]), int(labels_u8.shape[-1])
        update_long = self.frame_step - self.last_mem_step >= self.long_term_mem_gap
        with torch.cuda.stream(self.stream):
            s = self._s()
            new_slots = [-1] * B
            if update_long:
                self.last_mem_step = self.frame_step
                new_slots = [rt.free[c].pop(0) for c in range(B)]
                rt.upload_append_slots(new_slots, s)
            self._run(f'upd{int(update_long)}_{labels_u8.data_ptr()}', rt.prog_id_emb(labels_u8, hs, ws) + rt.prog_update(update_long))
            if not update_long:
                return
            for c in range(B):
                rt.slots[c].append(new_slots[c])
                self._indexes[c].append(self.frame_step)
            n_keep = self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN
            if rt.T > n_keep:
                if not self._mass_valid:
                    raise RuntimeError('long_term_mem_gap changed between propagate and update: attention mass not recorded')
                Tp = self._T_at_propagate
                L = rt.L
                evs = []
                for c in range(B):
                    evs.append(ops.evict_scores(rt.logits[c * rt.M4:(c + 1) * rt.M4], rt.mass[c * L * Tp:(c + 1) * L * Tp], rt.scores[c],
                                                ldl=16, nc=rt.nc, keep=self.obj_nums[0], Hi=rt.H4, Wi=rt.W4, He=rt.H16, We=rt.W16, T=Tp))
                ops.run(evs, s)
                for c in range(B):
                    ops.copy_async(rt.scores_host[c], rt.scores[c], 4 * Tp)(s)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self._pending = (Tp, ev)
            else:
                rt.upload_chunks(s)

    def _resolve_pending(self):
        if self._pending is None:
            return
        Tp, ev = self._pending
        self._pending = None
        ev.synchronize()
        rt = self.rt
        for c in range(self.B):
            drop = self.policies[c].choose(rt.scores_host[c, :Tp].clone(), self._indexes[c])
            self.drop_trace[c].append(drop)
            rt.free[c].append(rt.slots[c].pop(drop))
            del self._indexes[c][drop]
        with torch.cuda.stream(self.stream):
            rt.upload_chunks(self._s())

    def synchronize(self):
        self.stream.synchronize()
