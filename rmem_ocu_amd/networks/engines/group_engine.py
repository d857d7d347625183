"""Throughput engine: B clips of equal length advance in lockstep through ONE set of launch lists.

``GroupEngine`` is the batched counterpart of AOTEngine/AOTInferEngine (aot_engine.py) for clips with <= 10 objects whose
label masks are fed back.  Per clip it keeps exactly the host state AOTEngine keeps -- bank slot order,
``long_memories_indexes``, the eviction policy's EMA scores and visit counts (networks/layers/transformer.py:338-411), the
frame of the last long-term update (aot_engine.py:338-343) -- while the frame counter is shared, because clips of one length
get one gap (managers/evaluator.py:330-335).  All device work goes through rmem_ocu_amd.group_runtime.GroupRuntime: one
launch per layer for the whole group.

Covered protocols (the reference's evaluator, managers/evaluator.py:385-523):
  * restricted banks (N = former + latter, eviction by the RMem policy) and UNBOUNDED banks (latter_mem_len = 9999,
    tools/eval.py:92): the bank then grows by one entry per gap up to the 32 rows of the kernel's key table;
  * a NEW OBJECT appearing mid-clip in some of the clips (evaluator.py:484-508): that clip's frame is re-added as a reference
    frame -- its bank restarts at one entry and its long-term update schedule restarts at that frame (aot_engine.py:318-323),
    so from then on the clips of a group hold banks of DIFFERENT lengths and append at different frames.  The launches are laid
    out for the longest bank; shorter ones are padded with empty key-table rows (include/rmem.h: key_count 0), and bank appends
    go through the per-clip destination table (negative = no append for this clip).
R50-DeAOTL models run through group_runtime_deaot.GroupRuntimeDeAOT (same protocols; the eviction policy's scores and visit
counts then move on EVERY long-term update, deaot_engine.py / transformer.py:880-892).
SwinB-AOTL models (cfg 5) run through the same GroupRuntime with encoder_batch.SwinBatchEncoder as the look-ahead encoder.
Clips with > 10 objects and multi-scale / flip testing run on the per-clip engines, which are the drop-in API.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch

from ... import ops
from ...group_runtime import GroupRuntime
from ...group_runtime_deaot import GroupRuntimeDeAOT
from ...runtime import MAX_CHUNKS, ClipRuntime
from .aot_engine import MemoryPolicy

F32 = torch.float32


class GroupEngine:
    def __init__(self, aot_model, clips: int, gpu_id: int = 0, long_term_mem_gap: int = 9999, lookahead: int = 4, streams=None):
        self.cfg = aot_model.cfg
        self.AOT = aot_model
        self.B = clips
        self.deaot = self.cfg.MODEL_VOS == 'deaot'
        self.policy_every_update = self.deaot      # DeAOTEngine.policy_every_update
        self.gpu_id = gpu_id
        self.device = torch.device('cuda', gpu_id)
        self.align_corners = self.cfg.MODEL_ALIGN_CORNERS
        self.max_obj_num = aot_model.max_obj_num
        self.long_term_mem_gap = long_term_mem_gap
        self.lookahead = lookahead
        # streams = (main, encoder) made by the caller: the runtime deals streams onto its (4) hardware queues in CREATION order, so a
        # caller with several engines decides which streams share a queue by the order it creates them in (bench.py)
        self.stream = streams[0] if streams else torch.cuda.Stream(self.device)
        # the look-ahead encoder of the NEXT batch of frames runs here, beside the propagation of the current batch (events order
        # the two: a batch is propagated after its encoder finished, a buffer is re-encoded after its last frame was decoded)
        self.enc_stream = streams[1] if streams else torch.cuda.Stream(self.device)
        self._enc_done = [torch.cuda.Event(), torch.cuda.Event()]
        self._enc_free = [None, None]
        self.use_graphs = True
        self.rt: Optional[GroupRuntime] = None
        self._side: Optional[ClipRuntime] = None
        self._graphs: Dict[str, ops.Graph] = {}
        self.restart_engine()

    # ------------------------------------------------------------------ state
    def restart_engine(self):
        self.frame_step = 0
        self.last_mem_step: List[int] = [-1] * self.B
        self.obj_nums = None
        self._indexes: List[List[int]] = [[] for _ in range(self.B)]
        self.policies = [MemoryPolicy() for _ in range(self.B)]
        self.drop_trace: List[List[int]] = [[] for _ in range(self.B)]
        self._pending = None
        self._mass_valid = False
        self._T_at_propagate = 0
        self._Tc_at_propagate: List[int] = [0] * self.B
        if self.rt is not None:
            self.rt.reset_bank()

    def long_memories_indexes(self, clip: int) -> List[int]:
        self._resolve_pending()
        return self._indexes[clip]

    def _s(self) -> int:
        return self.stream.cuda_stream

    @property
    def n_keep(self) -> int:
        return self.cfg.FORMER_MEM_LEN + self.cfg.LATTER_MEM_LEN

    def _ensure_runtime(self, H: int, W: int):
        n = self.n_keep
        # +1: a restricted bank holds N + 1 entries between append and eviction; unbounded: as many as the key table has rows
        slots = n + 1 if n < MAX_CHUNKS else MAX_CHUNKS
        if self.rt is None or (self.rt.H, self.rt.W) != (H, W) or self.rt.S != slots:
            cls = GroupRuntimeDeAOT if self.deaot else GroupRuntime
            self.rt = cls(self.AOT.packed(), (H, W), slots, self.device, self.B, self.cfg.MODEL_LSTT_NUM, self.align_corners,
                          self.max_obj_num + 1, self.lookahead)
            self.label_in = torch.empty(self.B, H, W, dtype=F32, device=self.device)
            self._graphs = {}
            self._side = None
        return self.rt

    def _run(self, key: str, prog: list, s: Optional[int] = None):
        s = self._s() if s is None else s
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
        self.last_mem_step = [self.frame_step] * B
        self.policies = [MemoryPolicy() for _ in range(B)]
        for c in range(B):
            self._indexes[c].append(self.frame_step)

    def add_reference_frame_for(self, clip: int, img: torch.Tensor, label_u8: torch.Tensor):
        """Mid-clip reference frame for ONE clip of the group (a new object's mask arrived, evaluator.py:484-508 ->
        aot_engine.py:675-702, 241-325): img fp32 [3, H, W] at the network size, label_u8 uint8 [Ho, Wo] (the merged label map;
        resized to the network size by nearest neighbour inside the one-hot kernel) at a FIXED address.  The frame runs through a
        single-clip runtime in reference mode (same kernels); its K / V become the clip's only bank entry and its short-term
        memory, the clip's long-term schedule restarts here, ``long_memories_indexes`` keeps growing (the reference's quirk, 323),
        the eviction policy's state is reset (init_memory, transformer.py:438-443)."""
        rt, c = self.rt, clip
        self._resolve_pending()
        if self._side is None:
            if self.deaot:
                from ...runtime_deaot import DeAOTRuntime as Side
            else:
                Side = ClipRuntime
            self._side = Side(self.AOT.packed(), (rt.H, rt.W), 1, self.device, self.cfg.MODEL_LSTT_NUM, self.align_corners,
                              self.max_obj_num + 1)
            self._side_img = torch.empty(3, rt.H, rt.W, dtype=F32, device=self.device)
        side = self._side
        hs, ws = int(label_u8.shape[-2]), int(label_u8.shape[-1])
        L = rt.L
        with torch.cuda.stream(self.stream):
            s = self._s()
            ops.copy_async(self._side_img, img.contiguous(), 3 * rt.H * rt.W * 4)(s)
            side.prepare_pos(s)
            side.reset_bank()
            side.slots.append(side.take_slot())
            side.upload_chunks(s)
            ops.run(side.prog_encode(self._side_img) + side.prog_id_emb(label_u8, hs, ws) + side.prog_lstt(True, 1, side.slots[0]), s)
            # the clip's bank := this frame only (aot_engine.py:322), short-term memory := this frame's (transformer.py:675-678)
            rt.free[c] = sorted(rt.free[c] + rt.slots[c])
            new = rt.free[c].pop(0)
            rt.slots[c] = [new]
            for i in range(rt.NL):
                nk, nv = L * rt.bank_kw * 2, L * rt.bank_vw * 2          # bytes of one bank entry's keys / values
                ops.copy_async(rt.bank_K[i][c * rt.S + new], side.bank_K[i][side.slots[0]], nk)(s)
                ops.copy_async(rt.bank_V[i][c * rt.S + new], side.bank_V[i][side.slots[0]], nv)(s)
                ops.copy_async(rt.short_K[i][c * L:(c + 1) * L], side.short_K[i], nk)(s)
                ops.copy_async(rt.short_V[i][c * L:(c + 1) * L], side.short_V[i], nv)(s)
            rt.upload_chunks(s)
        self.last_mem_step[c] = self.frame_step
        self.policies[c] = MemoryPolicy()
        self._indexes[c].append(self.frame_step)

    # ------------------------------------------------------------------ look-ahead encoder
    def encode_inputs(self, buf: int = 0) -> torch.Tensor:
        """fp32 [lookahead * B, 3, H, W] of look-ahead buffer ``buf``: frame e of clip c goes to row e * B + c.  Fill it on
        ``enc_stream`` (the previous encoder pass over this buffer reads it there)."""
        return self.rt.enc_bufs[buf].img_in

    def encode_ahead(self, buf: int = 0):
        """Encode the frames in encode_inputs(buf) on the side stream; propagate_to_labels(enc_slot = buf * lookahead + e)
        waits for it."""
        es = self.enc_stream
        if self._enc_free[buf] is not None:
            es.wait_event(self._enc_free[buf])          # the buffer's previous frames have been projected / decoded
        with torch.cuda.stream(es):
            self._run(f'encB{buf}', self.rt.enc_bufs[buf].prog(), es.cuda_stream)
            self._enc_done[buf].record(es)

    # ------------------------------------------------------------------ propagate (aot_engine.py:398-465 + evaluator.py:430-441)
    def _will_append(self, c: int) -> bool:
        return self.frame_step - self.last_mem_step[c] >= self.long_term_mem_gap

    def _mass_needed(self) -> bool:
        """The attention mass of layer 0 is only read by the eviction policy: needed iff the update after this propagation appends
        to some clip's bank and overflows it (DeAOT: appends at all)."""
        need = any(self._will_append(c) and (self.policy_every_update or len(self.rt.slots[c]) + 1 > self.n_keep)
                   for c in range(self.B))
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
            self._Tc_at_propagate = [len(sl) for sl in rt.slots]
            wm = self._mass_needed()
            pk = f'post_{labels_u8.data_ptr()}_{Ho}_{Wo}'
            if pk not in rt._prog:
                rt._prog[pk] = [ops.logits_post(rt.logits, ldl=16, nc=rt.nc, keep=keep, Hi=rt.H4, Wi=rt.W4, Ho=Ho, Wo=Wo,
                                                align_corners=self.align_corners, label_u8=labels_u8, images=B)]
            if enc_slot is None:
                ops.copy_async(rt.enc_now.img_in, imgs.contiguous(), B * 3 * rt.H * rt.W * 4)(self._s())
                prog = rt.prog_encode() + rt.prog_project(None) + rt.prog_lstt(False, T, wm) + rt.prog_decode(None) + rt._prog[pk]
            else:
                buf = enc_slot // rt.lookahead
                self.stream.wait_event(self._enc_done[buf])
                prog = rt.prog_project(enc_slot) + rt.prog_lstt(False, T, wm) + rt.prog_decode(enc_slot) + rt._prog[pk]
            self._run(f'prop{T}{int(wm)}e{enc_slot}_{labels_u8.data_ptr()}', prog)
            if enc_slot is not None:
                ev = self._enc_free[buf] or torch.cuda.Event()
                ev.record(self.stream)
                self._enc_free[buf] = ev

    # ------------------------------------------------------------------ memory update (aot_engine.py:327-369)
    def update_from_labels(self, labels_u8: torch.Tensor, skip: Iterable[int] = ()):
        """labels_u8: uint8 [B, Ho, Wo] argmax labels at the output size (nearest-resized to the network size on the device).
        skip: clips whose memory is re-initialised right after by add_reference_frame_for (no bank append for them)."""
        rt, B = self.rt, self.B
        hs, ws = int(labels_u8.shape[-2]), int(labels_u8.shape[-1])
        skip = set(skip)
        appends = [self._will_append(c) and c not in skip for c in range(B)]
        with torch.cuda.stream(self.stream):
            s = self._s()
            new_slots = [-1] * B
            if any(appends):
                for c in range(B):
                    if appends[c]:
                        if not rt.free[c]:
                            raise ops.RmemError(f'clip {c}: the memory bank outgrew the {rt.S} slots of the group runtime '
                                                f'(unbounded banks are limited by the {MAX_CHUNKS}-row key table)')
                        new_slots[c] = rt.free[c].pop(0)
                rt.upload_append_slots(new_slots, s)
            self._run(f'upd{int(any(appends))}_{labels_u8.data_ptr()}', rt.prog_id_emb(labels_u8, hs, ws) + rt.prog_update(any(appends)))
            if not any(appends):
                return
            scored, over = [], set()            # clips whose policy state moves / whose bank overflows
            for c in range(B):
                if appends[c]:
                    self.last_mem_step[c] = self.frame_step
                    rt.slots[c].append(new_slots[c])
                    self._indexes[c].append(self.frame_step)
                    if len(rt.slots[c]) > self.n_keep:
                        over.add(c)
                    if c in over or self.policy_every_update:
                        scored.append(c)
            if scored:
                if not self._mass_valid:
                    raise RuntimeError('long_term_mem_gap changed between propagate and update: attention mass not recorded')
                Tp, L = self._T_at_propagate, rt.L
                ops.run([ops.evict_scores(rt.logits[c * rt.M4:(c + 1) * rt.M4], rt.mass[c * L * Tp:(c + 1) * L * Tp], rt.scores[c],
                                          ldl=16, nc=rt.nc, keep=self.obj_nums[0], Hi=rt.H4, Wi=rt.W4, He=rt.H16, We=rt.W16, T=Tp)
                         for c in scored], s)
                for c in scored:
                    ops.copy_async(rt.scores_host[c], rt.scores[c], 4 * Tp)(s)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self._pending = (ev, scored, over, list(self._Tc_at_propagate))
            else:
                rt.upload_chunks(s)

    def _resolve_pending(self):
        if self._pending is None:
            return
        ev, scored, over, tc = self._pending
        self._pending = None
        ev.synchronize()
        rt = self.rt
        for c in scored:
            drop = self.policies[c].choose(rt.scores_host[c, :tc[c]].clone(), self._indexes[c])
            if c not in over:                 # DeAOT: the scores moved, nothing is dropped yet
                continue
            self.drop_trace[c].append(drop)
            rt.free[c].append(rt.slots[c].pop(drop))
            del self._indexes[c][drop]
        with torch.cuda.stream(self.stream):
            rt.upload_chunks(self._s())

    def synchronize(self):
        self.stream.synchronize()
        self.enc_stream.synchronize()
