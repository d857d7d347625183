"""Clip-parallel driver: many clips in flight per GPU, one process per GPU.

Mirrors the reference's evaluation protocol (managers/evaluator.py:330-335, 385-441, 509-523:
per-sequence gap = max(round(n/30), 5), reference frame, then propagate -> argmax -> update per
frame) and its multi-GPU scheme (tools/eval.py:137-143, evaluator.py:276-295: one worker per GPU
draining a queue of sequences; no collective on the data path, one final gather of statistics).
Frames inside a clip are strictly sequential (short-term memory recurrence + mask feedback), so
a GPU is filled by interleaving independent clips: each clip owns an engine, a HIP stream and its
hipGraphs; the host only enqueues.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops


def shard_clips(num_clips: int, rank: int, world: int, lengths: Optional[Sequence[int]] = None) -> List[int]:
    """Static longest-first assignment of clip ids to ranks (the reference pops a shared queue; with
    equal hardware a greedy longest-first split gives the same balance without a host-side queue)."""
    ids = list(range(num_clips))
    if lengths is not None:
        ids.sort(key=lambda i: (-lengths[i], i))
    loads = [0] * world
    mine: List[int] = []
    for i in ids:
        r = min(range(world), key=lambda j: (loads[j], j))
        loads[r] += lengths[i] if lengths is not None else 1
        if r == rank:
            mine.append(i)
    return mine


def group_units(lengths: Sequence[int], group: int) -> List[List[int]]:
    """Work units of a clip list for engines that advance ``group`` clips in lockstep: clips of EQUAL length (equal length
    => equal gap, managers/evaluator.py:330-335) are bundled ``group`` at a time, longest first; a length whose clip count is
    not a multiple of ``group`` leaves one shorter unit (run by a smaller group or the per-clip engine)."""
    by_len = {}
    for i, n in enumerate(lengths):
        by_len.setdefault(int(n), []).append(i)
    units = []
    for n in sorted(by_len, reverse=True):
        ids = by_len[n]
        units += [ids[k:k + group] for k in range(0, len(ids), group)]
    return units


class ClipFeeder:
    """Hands the work units of a job's clip list to this rank -- the counterpart of the reference's sequence queue
    (tools/eval.py:137-143 spawns one worker per GPU; managers/evaluator.py:276-295 has every worker pop sequences from one
    shared mp.Queue until it is empty).  No data-path collective is involved in either mode:

      'queue'  (default for world > 1) a job-wide ticket counter on the process group's TCPStore (``store.add``): whichever
               rank has a free slot takes the next unit, longest first, so a slow rank or a skewed list costs at most one unit
               of tail (host-side work stealing; one ~50 us store round trip per CLIP, never per frame);
      'static' the longest-first greedy split of ``shard_clips`` computed identically on every rank (no store traffic).

    ``cyclic``: after the last unit the list starts over (bench.py measures a fixed-length window of a long job).

    ``store``: any c10d store shared by the ranks (bench.py passes the TCPStore its process group was initialised with); the
    feeder works inside its own PrefixStore namespace, ``<key>/<n>`` for the n-th feeder this process created -- every rank
    creates its feeders in the same order, so the n-th feeders of all ranks share one ticket counter that starts at zero, and
    a second job in the same process group does not inherit the first one's tickets."""

    _instances = 0

    def __init__(self, lengths: Sequence[int], rank: int = 0, world: int = 1, group: int = 1, mode: Optional[str] = None,
                 store=None, cyclic: bool = False, key: str = 'rmem_clip_queue'):
        self.lengths = [int(n) for n in lengths]
        self.units = group_units(self.lengths, group)
        self.rank, self.world, self.cyclic = rank, world, cyclic
        self.key = f'{key}/{ClipFeeder._instances}'
        ClipFeeder._instances += 1
        self.mode = mode or ('queue' if world > 1 else 'static')
        if self.mode not in ('queue', 'static'):
            raise ValueError(f'ClipFeeder mode {self.mode!r}: expected "queue" or "static"')
        if self.mode == 'queue' and world > 1:
            if store is None:
                raise ValueError('ClipFeeder: the job-wide queue needs the c10d store the ranks share (store=...)')
            import torch.distributed as dist
            self.store = dist.PrefixStore(self.key, store)
        else:
            self.store = None
        unit_len = [self.lengths[u[0]] * len(u) for u in self.units]
        self._mine = shard_clips(len(self.units), rank, world, unit_len)      # static order (also the world == 1 order)
        self._taken = 0
        self.history: List[int] = []         # unit indexes this rank ran, in order

    def next_unit(self) -> Optional[List[int]]:
        """Clip ids of the next unit for this rank, or None when the list is drained."""
        n = len(self.units)
        if self.store is not None:
            t = int(self.store.add('ticket', 1)) - 1                  # job-wide ticket
            if t >= n and not self.cyclic:
                return None
            u = t % n
        else:
            if not self._mine or (self._taken >= len(self._mine) and not self.cyclic):
                return None
            u = self._mine[self._taken % len(self._mine)]
            self._taken += 1
        self.history.append(u)
        return self.units[u]


def open_job_store(rank: int, world: int):
    """The c10d store of this job, opened explicitly from MASTER_ADDR / MASTER_PORT (``env://`` rendezvous: under
    torch.distributed.run it is a client of the launcher's TCPStore, otherwise rank 0 hosts it).  bench.py hands the SAME store
    to ``init_process_group(store=...)`` and to ClipFeeder, so nothing reaches into the process group's internals."""
    import torch.distributed as dist
    store, _, _ = next(iter(dist.rendezvous('env://', rank, world)))
    return store


DRAIN = 1 << 62


def pump(slots: Sequence, start_fn, steps: int, frames_per_step: int = 1) -> int:
    """Advance the slots round-robin until ``steps`` propagated frames have been enqueued (one slot step = ``frames_per_step``
    frames: the clips of a group move together).  A slot whose clip has ended takes the next unit of the job through
    ``start_fn(slot)`` (reference frames: executed, not counted); start_fn returns False when the list is drained, the slot then
    idles.  Returns the frames actually enqueued (< steps only if every slot ran dry: ``steps = DRAIN`` runs the job to its end,
    bench.py --drain).  Host-side only: nothing here waits for the GPU."""
    done, j, idle = 0, 0, 0
    n = len(slots)
    while done < steps and idle < n:
        s = slots[j % n]
        j += 1
        if s.done and start_fn(s) is False:
            idle += 1
            continue
        if s.done:                      # a unit of single-frame clips: nothing to propagate
            idle = 0
            continue
        idle = 0
        s.step()
        done += frames_per_step
    return done


def gather_stats(frames: float, seconds: float, checksum: float, dist, rank: int, world: int, device):
    """The one exchange of the clip-parallel design (reference: info_queue, managers/evaluator.py:589-613):
    every rank sends (frames, seconds, checksum) to rank 0, which returns (sum frames, max seconds, sum checksum)."""
    stats = torch.tensor([frames, seconds, checksum], dtype=torch.float64, device=device)
    if dist is None or world == 1:
        return float(stats[0]), float(stats[1]), float(stats[2])
    gathered = [torch.zeros_like(stats) for _ in range(world)] if rank == 0 else None
    dist.gather(stats, gathered, dst=0)
    if rank != 0:
        return None
    return (sum(float(g[0]) for g in gathered), max(float(g[1]) for g in gathered), sum(float(g[2]) for g in gathered))


class ClipSlot:
    """One clip in flight: an engine, its frames on the device, its output label buffer."""

    def __init__(self, engine, out_hw, device, lookahead: int = 1):
        self.engine = engine
        self.lookahead = lookahead          # > 1: the encoder runs this many frames ahead, one launch per layer for all of them
        self.labels: Optional[torch.Tensor] = None
        self.cur_label = torch.zeros(out_hw[0], out_hw[1], dtype=torch.uint8, device=device)   # fixed address (graph-captured)
        self.out_hw = out_hw
        self.device = device
        self.frames = None
        self.cursor = 0
        self.done = True
        self.frames_encoded = 0             # frames that went through the encoder (reference frames + look-ahead batches)

    def start(self, frames: torch.Tensor, first_mask: torch.Tensor, num_objs: int):
        """frames [n,3,H,W] fp32 device at network size, or decoded uint8 RGB [n,Hs,Ws,3] in PINNED HOST memory (then every
        frame crosses PCIe as uint8 and is resized + normalised on the device, rmem_ingest_rgb8);
        first_mask [1,1,H,W] at network size."""
        n = frames.shape[0]
        self.frames = frames
        self.host_u8 = frames.dtype == torch.uint8
        if self.labels is None or self.labels.shape[0] < n:
            self.labels = torch.zeros(n, self.out_hw[0], self.out_hw[1], dtype=torch.uint8, device=self.device)
        eng = self.engine
        eng.restart_engine()
        eng.long_term_mem_gap = max(int(round(n / 30)), 5)      # evaluator.py:330-335
        if self.host_u8:
            if not frames.is_pinned():
                raise ValueError('uint8 host frames must be in pinned memory')
            H, W = int(first_mask.shape[-2]), int(first_mask.shape[-1])
            hs, ws = int(frames.shape[1]), int(frames.shape[2])
            if getattr(self, '_stage', None) is None or self._stage.shape[1:3] != (hs, ws):
                self._stage = torch.empty(max(self.lookahead, 1), hs, ws, 3, dtype=torch.uint8, device=self.device)
                self._first = torch.empty(1, 3, H, W, dtype=torch.float32, device=self.device)
            self._net_hw = (H, W)
            cur = torch.cuda.current_stream(self.device)
            # the engine's own stream may still be running the previous clip's last frames, which read _stage / _first:
            # order this clip's first-frame copy + ingest behind them (a device-side wait, no host stall)
            for e in eng.aot_engines + getattr(eng, '_pool', []):
                cur.wait_stream(e.stream)
            ops.copy_async(self._stage[0], frames[0], hs * ws * 3)(cur.cuda_stream)
            ops.run(ops.ingest_rgb8(self._stage[0], Hs=hs, Ws=ws, Hd=H, Wd=W, out_chw=self._first[0]), cur.cuda_stream)
            cur.synchronize()        # once per clip: the engine works on its own stream
            eng.add_reference_frame(self._first, first_mask, obj_nums=[num_objs], frame_step=0)
        else:
            eng.add_reference_frame(frames[0:1], first_mask, obj_nums=[num_objs], frame_step=0)
        self.frames_encoded += 1
        self.cursor = 1
        self.done = n <= 1

    def _ingest_group(self, i: int):
        """Host -> device copy of the next look-ahead group of uint8 frames and their resize + normalise into the encoder's
        input buffer, all on the clip's stream."""
        eng = self.engine
        s = eng.aot_engines[0].stream.cuda_stream
        la = max(self.lookahead, 1)
        m = min(la, self.frames.shape[0] - i)
        hs, ws = int(self.frames.shape[1]), int(self.frames.shape[2])
        H, W = self._net_hw
        ops.copy_async(self._stage, self.frames[i:i + m], m * hs * ws * 3)(s)
        dst = eng.encode_inputs(la) if la > 1 else self._first
        ops.run([ops.ingest_rgb8(self._stage[b], Hs=hs, Ws=ws, Hd=H, Wd=W, out_chw=dst[b]) for b in range(m)], s)

    def step(self):
        """Propagate one frame and update the memory with the predicted labels (all asynchronous)."""
        i = self.cursor
        if self.lookahead > 1:
            e = (i - 1) % self.lookahead
            if e == 0:
                self.frames_encoded += min(self.lookahead, self.frames.shape[0] - i)
                if self.host_u8:
                    self._ingest_group(i)
                    self.engine.encode_ahead(None, self.lookahead)
                else:
                    self.engine.encode_ahead(self.frames[i:i + self.lookahead], self.lookahead)
            self.engine.propagate_to_label(None, self.cur_label, enc_slot=e)
        elif self.host_u8:
            self.frames_encoded += 1
            self._ingest_group(i)
            self.engine.propagate_to_label(self._first, self.cur_label)
        else:
            self.frames_encoded += 1
            self.engine.propagate_to_label(self.frames[i:i + 1], self.cur_label)
        self.engine.update_memory_from_label_u8(self.cur_label)
        # the clip's delivered masks stay on the device
        ops.copy_async(self.labels[i], self.cur_label, self.cur_label.numel())(self.engine.aot_engines[0].stream.cuda_stream)
        self.cursor += 1
        self.done = self.cursor >= self.frames.shape[0]


class GroupSlot:
    """B clips of equal length in flight on one GroupEngine (networks/engines/group_engine.py): same protocol as ClipSlot, one
    launch per layer for the whole group.  With an encoder look-ahead of n frames the frames are encoded in batches of n per
    clip: batch k + 1 is copied in and encoded on the engine's side stream while the frames of batch k are propagated (two
    look-ahead buffers, alternating; look-ahead slot = buffer * n + frame)."""

    def __init__(self, engine, out_hw, device):
        self.engine = engine
        self.B = engine.B
        self.out_hw = out_hw
        self.device = device
        self.cur_label = torch.zeros(self.B, out_hw[0], out_hw[1], dtype=torch.uint8, device=device)   # fixed address (graph-captured)
        self.labels: Optional[torch.Tensor] = None
        self.frames = None
        self.cursor = 0
        self.done = True
        self.frames_encoded = 0             # frames that went through the encoder (reference frames + look-ahead batches)

    def start(self, frames: Sequence[torch.Tensor], first_masks: Sequence[torch.Tensor], num_objs: int, new_objects=None):
        """frames: B tensors [n, 3, H, W] fp32 device (equal n), or B uint8 [n, Hs, Ws, 3] tensors in PINNED HOST memory (every
        frame then crosses PCIe as uint8 and is resized + normalised on the device); first_masks: B tensors [1, 1, H, W] at the
        network size.  new_objects: {clip index: (frame index, uint8 [Ho, Wo] device map: the new object's label on its pixels, 0
        elsewhere)} -- the evaluator's protocol for an object that appears mid-clip (managers/evaluator.py:484-508): the frame is
        propagated, the new label is laid over the prediction and the frame is re-added as a reference frame for that clip."""
        assert len(frames) == self.B and len({int(f.shape[0]) for f in frames}) == 1
        n = int(frames[0].shape[0])
        if getattr(self, '_frames_by_pointer', False):
            self.engine.enc_stream.synchronize()  # queued encoder launches read the previous clips' frames in place: let go of them after
        self.frames = list(frames)
        self.host_u8 = frames[0].dtype == torch.uint8
        self.new_objects = dict(new_objects or {})
        if self.new_objects and self.host_u8:
            raise ValueError('new_objects: frames must be fp32 device tensors at the network size')
        if self.labels is None or self.labels.shape[1] < n:
            self.labels = torch.zeros(self.B, n, self.out_hw[0], self.out_hw[1], dtype=torch.uint8, device=self.device)
        eng = self.engine
        self._frames_by_pointer = not self.host_u8 and eng.lookahead > 1
        eng.restart_engine()
        eng.long_term_mem_gap = max(int(round(n / 30)), 5)      # evaluator.py:330-335
        H, W = int(first_masks[0].shape[-2]), int(first_masks[0].shape[-1])
        with torch.cuda.stream(eng.stream):
            s = eng.stream.cuda_stream
            if self.host_u8:
                hs, ws = int(frames[0].shape[1]), int(frames[0].shape[2])
                la = max(eng.lookahead, 1)
                if getattr(self, '_stage', None) is None or tuple(self._stage.shape[1:3]) != (hs, ws):
                    self._stage = torch.empty(la * self.B, hs, ws, 3, dtype=torch.uint8, device=self.device)
                    self._first = torch.empty(self.B, 3, H, W, dtype=torch.float32, device=self.device)
                for c in range(self.B):
                    if not frames[c].is_pinned():
                        raise ValueError('uint8 host frames must be in pinned memory')
                    ops.copy_async(self._stage[c], frames[c][0], hs * ws * 3)(s)
                ops.run([ops.ingest_rgb8(self._stage[c], Hs=hs, Ws=ws, Hd=H, Wd=W, out_chw=self._first[c]) for c in range(self.B)], s)
                imgs = self._first
            else:
                imgs = torch.cat([f[0:1] for f in frames], 0)
            masks = torch.cat([m.reshape(1, 1, m.shape[-2], m.shape[-1]).float() for m in first_masks], 0)
        eng.add_reference_frames(imgs, masks, num_objs)
        self.frames_encoded += self.B
        self.cursor = 1
        self.done = n <= 1
        if eng.lookahead > 1 and n > 1:
            self._kick_encoder(0, 1)                            # frames 1 .. lookahead: batch 0

    def _kick_encoder(self, buf: int, i: int):
        """Encode frames i .. i + lookahead - 1 of every clip into look-ahead buffer ``buf`` on the engine's side stream."""
        eng = self.engine
        m = min(eng.lookahead, self.frames[0].shape[0] - i)
        self.frames_encoded += self.B * m
        stage = None
        if self.host_u8:
            if getattr(self, '_stage_la', None) is None or tuple(self._stage_la.shape[1:3]) != tuple(self.frames[0].shape[1:3]):
                hs, ws = int(self.frames[0].shape[1]), int(self.frames[0].shape[2])
                self._stage_la = torch.empty(eng.lookahead * self.B, hs, ws, 3, dtype=torch.uint8, device=self.device)
            stage = self._stage_la
        enc = eng.rt.enc_bufs[buf]
        if self.host_u8:
            enc.point_at_img_in(eng.enc_stream.cuda_stream)
            self._fill_encoder_inputs(eng.encode_inputs(buf), i, m, stream=eng.enc_stream, stage=stage)
        else:
            # the encoder reads the frames where the clips are (device table of pointers, row k * B + c = frame i + k of clip c): no
            # staging copy; rows beyond the clip's end name its last frame again (encoded, never used)
            last = self.frames[0].shape[0] - 1
            enc.set_frames([self.frames[c][min(i + k, last)] for k in range(eng.lookahead) for c in range(self.B)], eng.enc_stream.cuda_stream)
        eng.encode_ahead(buf)

    def _fill_encoder_inputs(self, dst: torch.Tensor, i: int, m: int, stream=None, stage=None):
        """frames i .. i + m - 1 of every clip into rows k * B + c of dst ([., 3, H, W] fp32) on ``stream`` (default: the engine's
        main stream), uint8 host frames through the staging buffer ``stage``."""
        eng, B = self.engine, self.B
        s = (stream or eng.stream).cuda_stream
        stage = getattr(self, '_stage', None) if stage is None else stage
        if self.host_u8:
            hs, ws = int(self.frames[0].shape[1]), int(self.frames[0].shape[2])
            H, W = int(dst.shape[-2]), int(dst.shape[-1])
            for c in range(B):
                ops.copy_async(stage[c * m:(c + 1) * m], self.frames[c][i:i + m], m * hs * ws * 3)(s)
            ops.run([ops.ingest_rgb8(stage[c * m + k], Hs=hs, Ws=ws, Hd=H, Wd=W, out_chw=dst[k * B + c]) for c in range(B)
                     for k in range(m)], s)
        else:
            fb = dst[0].numel() * 4
            for c in range(B):
                for k in range(m):
                    ops.copy_async(dst[k * B + c], self.frames[c][i + k], fb)(s)

    def step(self, feed: Optional[torch.Tensor] = None):
        """Propagate frame ``cursor`` of every clip, deliver its labels and update the memories.  feed: uint8 [B, Ho, Wo] device
        labels that go into the memory update INSTEAD of the prediction (the delivered labels stay the prediction): given masks,
        e.g. the reference's own in the parity tests, so that every frame is an independent comparison."""
        eng, B, i = self.engine, self.B, self.cursor
        la = eng.lookahead
        s = eng.stream.cuda_stream
        if la > 1:
            b, e = divmod(i - 1, la)
            if e == 0 and i + la < self.frames[0].shape[0]:
                self._kick_encoder((b + 1) % 2, i + la)         # the NEXT batch, on the side stream, beside this batch's frames
            eng.propagate_to_labels(self.cur_label, enc_slot=(b % 2) * la + e)
        else:
            self.frames_encoded += B
            if self.host_u8:
                self._fill_encoder_inputs(self._first, i, 1)
                imgs = self._first
            else:
                with torch.cuda.stream(eng.stream):
                    imgs = torch.cat([f[i:i + 1] for f in self.frames], 0)
            eng.propagate_to_labels(self.cur_label, imgs=imgs)
        inject = [c for c, (fi, _) in self.new_objects.items() if fi == i]
        nb = self.cur_label[0].numel()                # frame i of every clip's label stack: one pitched copy
        if feed is not None:                          # deliver the prediction, then continue from the given labels
            ops.copy2d_async(self.labels.view(-1)[i * nb:], self.labels.shape[1] * nb, self.cur_label, nb, nb, B)(s)
            ops.copy_async(self.cur_label, feed.contiguous(), B * nb)(s)
        if inject:
            with torch.cuda.stream(eng.stream):
                for c in inject:                      # the new object's label over the prediction (evaluator.py:484-497)
                    new = self.new_objects[c][1]
                    self.cur_label[c].copy_(torch.where(new > 0, new, self.cur_label[c]))
        eng.update_from_labels(self.cur_label, skip=inject)
        for c in inject:
            eng.add_reference_frame_for(c, self.frames[c][i], self.cur_label[c])
        if feed is None:
            ops.copy2d_async(self.labels.view(-1)[i * nb:], self.labels.shape[1] * nb, self.cur_label, nb, nb, B)(s)
        self.cursor += 1
        self.done = self.cursor >= self.frames[0].shape[0]
