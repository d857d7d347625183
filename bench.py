#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X memory-reading VOS engine on BASELINE.json's cfg 2.

    python bench.py --gpus N --steps K --warmup W

N > 1 without WORLD_SIZE in the environment: this process starts N fresh ranks itself (``python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` as a child, before anything here touches the GPU -- the way the
reference's tools/eval.py:137-143 spawns one worker per GPU), relays rank 0's JSON line and exits with the child's code.
Launched by torch.distributed.run (WORLD_SIZE set) it is one rank.

Workload (config.workload = "davis17_480p_r50_N8"): a job of 64 synthetic 480x854 clips per GPU, 80 frames each, at network
size 481x849 (HW = 1674 tokens), ResNet-50 + 3-layer LSTT + FPN, memory bank N = 8 (1 + 7), per-clip
gap = max(round(80/30), 5) = 5, 3 objects, bf16 operands / fp32 accumulation.  Other workloads (--workload): the DeAOT model, the
cfg-3 protocol (720p, a new object arriving at frame 15), cfg 4 (1080p, 600-frame clips, restricted vs unbounded bank), cfg 5.

The unit of work is one propagated frame of one clip: match-propagate + argmax + memory update -- the reference's own FPS unit
(managers/evaluator.py:399-404, 525-535) -- and `value` is always frames / second.  A *step* (--steps K, --warmup W) is one pass
of the engine over its batch: ONE FRAME OF EVERY CLIP IN FLIGHT on the GPU (24 clips = 3 groups of 8: config.frames_per_step
= 24), so --steps 20 times 480 frames.  (--step-unit frame restores the rounds 1-2 reading, one frame of one clip per step:
--steps 20 was then five 4-clip group steps started from an idle GPU, i.e. a fill / drain transient 12-15 % below the rate the same
build sustains, swinging +-5 % with what the window happened to contain; DESIGN.md section 6 quotes both.)
Clips are independent; the job's clip list is handed to the ranks by rmem_ocu_amd.clip_runner.ClipFeeder -- a job-wide ticket
queue on the job's TCPStore (the reference's shared sequence queue, managers/evaluator.py:276-295) or, with --feeder static, a
longest-first split -- and every rank keeps 24 clips in flight as 3 groups of 8 clips that advance in lockstep on one
GroupEngine each (one launch per layer for the 8 clips; --clips-per-group 1 selects the per-clip engines of the drop-in API),
every group on its own HIP stream with its own hipGraphs.  Ranks never exchange data on the hot path (weak scaling: per-GPU
work is fixed); the only collectives are the barriers around the timed region and one final gather of (frames, seconds,
checksum) to rank 0.  The timed region is a window of K steps per rank out of that job (the list is cyclic, a window never
runs dry); --drain instead runs the whole job once from a non-cyclic list (value = all ranks' frames / the slowest rank's
seconds: tail and imbalance included).  Reference frames that fall inside the timed region are executed but not counted.
Inside a clip the ResNet-50 encoder runs 2 frames ahead of the LSTT on a side stream (frames do not depend on each other before
the memory read): one launch per encoder layer covers 2 frames x 8 clips, every frame is still encoded exactly once
(config.encoder_lookahead; config.frames_encoded_in_timed_region counts the frames whose encoder was enqueued inside the window:
it equals config.frames_counted_in_timed_region up to one look-ahead batch per group -- the encoder pipeline is equally far
ahead at both ends of the window, so no encoder work is moved out of it).  Inputs are resident in HBM when the timed region starts.

The single JSON line also carries
  roofline     -- the dominant kernel (the long-term memory read, rmem_mem_read_attn_clips): AFTER the timed region (so
                  it cannot perturb `value`, and independent of --steps) the layer-0 memory read of one group is
                  launched --roofline-launches times (default 32) at bank size T = 8 (the steady state of the
                  clip) alone on the GPU, directly (events cannot sit inside a replayed graph), each launch bracketed by
                  HIP events on the launch stream.  achieved = algorithmic FLOPs 4*HW*(T*HW)*256 per clip-launch /
                  mean launch duration; these launches use the symbol k_attn_partial<true, true>, so `rocprofv3
                  --kernel-trace --stats` of the same command reports exactly them under that name; peak = 2.5 PFLOP/s
                  dense bf16.  `t_mix` says which bank sizes were timed.  `traffic` (HBM bytes per launch from separate
                  rocprofv3 --pmc passes) is only reported while the committed PMC summary was measured on the attention.hip that is
                  built now (sha256 match), else null.
  cpu_baseline -- oracle/ref_cpu.py (fp32 port of the reference path) timed on this host's cores on a
                  bounded sample of the same workload (rank 0, N = 1 only).
  check        -- (with cpu_baseline) the masks of the timed path against the checker: the first clip group re-runs the clip the
                  oracle just ran, same protocol (gap 1, 16 frames: the bank fills and evicts), free-running; per-pixel label
                  agreement with the oracle's masks (the line is withheld below 0.97; near-ties of the synthetic weights flip ~0.3 %).
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0
# --workload: the default is BASELINE.json's cfg 2 (the line the driver records); the others are extra measurements
WORKLOADS = {
    'davis17_480p_r50_N8': dict(model='r50_aotl', video=(480, 854), lengths=(80,), clips_per_gpu=64, objs=3, former=1, latter=7, net=None),
    # the model the reference's shipped eval_vost.sh runs: R50-DeAOTL, bank 1 + 8 (configs/models/r50_deaotl.py:8-9), cfg-2 geometry
    'davis17_480p_r50deaot_N9': dict(model='r50_deaotl', video=(480, 854), lengths=(80,), clips_per_gpu=64, objs=3, former=1, latter=8, net=None),
    # cfg 5 geometry: 720p, Swin-B, bank N = 12 (1 + 11), align_corners False -> network size = video size (multiple of 16)
    'lvos_720p_swinb_N12': dict(model='swinb_aotl', video=(720, 1280), lengths=(150,), clips_per_gpu=4, objs=2, former=1, latter=11, net=(720, 1280),
                                dtype='fp16'),
    # cfg 2 geometry with a skewed clip list (mixed lengths): exercises the feeder's length buckets and queue
    'davis17_480p_r50_N8_mixed': dict(model='r50_aotl', video=(480, 854), lengths=(100, 80, 60, 40), clips_per_gpu=64, objs=3, former=1, latter=7, net=None),
    # cfg 3 geometry and protocol: 720p -> network size 577x1041 (HW = 2442), 36-frame clips, a NEW OBJECT's mask arrives at frame
    # 15 of every clip (managers/evaluator.py:484-508: the frame is re-added as a reference frame, the bank restarts at one entry)
    'ytvos_720p_r50_N8_inject': dict(model='r50_aotl', video=(720, 1280), lengths=(36,), clips_per_gpu=64, objs=2, former=1, latter=7, net=None,
                                     inject_at=15),
    # cfg 4 geometry: 1080p -> 577x1041, 600-frame clips (gap = 20), one object; restricted bank N = 8 against the unbounded bank
    # (tools/eval.py:92, latter_mem_len = 9999: T grows to 30) -- the stress case of the memory-read kernel
    'vost_1080p_r50_N8': dict(model='r50_aotl', video=(1080, 1920), lengths=(600,), clips_per_gpu=24, objs=1, former=1, latter=7, net=None,
                              distinct_frames=100),
    'vost_1080p_r50_unbounded': dict(model='r50_aotl', video=(1080, 1920), lengths=(600,), clips_per_gpu=24, objs=1, former=1, latter=9999,
                                     net=None, distinct_frames=100, roofline_T=(8, 30)),
}


def cpu_baseline(frames, mask, video_hw, n_timed=8):
    """Oracle (CPU port of the reference path) on a bounded sample: bank filled to N = 8 with gap 1 over 8
    untimed frames, then n_timed propagated frames at T = 8 are timed (propagate + update)."""
    from oracle import ref_cpu as O
    from rmem_ocu_amd.weights import synth_state_dict
    import torch.nn.functional as F
    threads = int(os.environ.get('RMEM_CPU_THREADS', min(16, os.cpu_count() or 1)))   # the 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(threads)
    eng = O.OracleEngine(synth_state_dict(0), 1, 7, 1)
    with torch.no_grad():
        eng.add_reference_frame(frames[0:1], mask, 0)
        t0 = None
        labels = []
        for i in range(1, 9 + n_timed):
            if i == 9:
                t0 = time.time()
            logit = eng.match_propogate_one_frame(frames[i:i + 1], video_hw)
            label = torch.argmax(torch.softmax(logit, 1), 1, keepdim=True).float()
            eng.update_memory(F.interpolate(label, size=eng.input_size_2d, mode='nearest'))
            labels.append(label[0, 0].to(torch.uint8))
        dt = time.time() - t0
    return {'value': round(n_timed / dt, 4), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n_timed} propagated frames at 481x849, bank T=8 (steady state of the 80-frame clip), fp32, after 9 untimed frames'}, torch.stack(labels), list(eng.long_memories_indexes)


def check_against_oracle(slot, frames_dev, mask_dev, num_objs, oracle_labels, oracle_indexes):
    """The timed path against the checker: the first group slot runs the clip the CPU oracle just ran (every clip of the group is that
    clip), same protocol (gap 1: the bank fills to N = 8 and evicts), free-running, and its delivered masks are compared with the
    oracle's frame by frame.  Outside every timed region; a line whose masks disagree is not printed."""
    G, n = slot.B, int(oracle_labels.shape[0])
    slot.start([frames_dev[:n + 1]] * G, [mask_dev] * G, num_objs)
    slot.engine.long_term_mem_gap = 1
    while not slot.done:
        slot.step()
    slot.engine.synchronize()
    got = slot.labels[:, 1:n + 1].cpu()
    agree = [float((got[c] == oracle_labels).float().mean()) for c in range(G)]
    per_frame = (got[0] == oracle_labels).float().flatten(1).mean(1)
    if min(agree) < 0.97 or any(not torch.equal(got[0], got[c]) for c in range(1, G)):
        raise SystemExit(f'bench.py: the HIP path disagrees with the oracle (label agreement per clip {agree})')
    return {'against': 'oracle/ref_cpu.py (fp32 CPU port), same clip and protocol, free-running', 'frames': n, 'clips': G,
            'label_agreement': round(min(agree), 5), 'worst_frame': round(float(per_frame.min()), 5),
            'bank_indexes': list(slot.engine.long_memories_indexes(0)), 'bank_indexes_equal_the_oracles': list(slot.engine.long_memories_indexes(0)) == oracle_indexes}


def attention_source_sha256() -> str:
    with open(os.path.join(ROOT, 'rmem_ocu_amd', 'csrc', 'attention.hip'), 'rb') as f:
        return hashlib.sha256(f.read()).hexdigest()


def pmc_traffic(clips_per_launch=1):
    """HBM bytes per T = 8 launch of the memory-read kernel from the newest committed PMC summary (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 FETCH correction applied; bench.py cannot collect counters itself).  A summary is
    only valid for the kernel source it was measured on: it names attention.hip's sha256, and anything else gives None."""
    name = {1: 'attn_pmc.json', 4: 'attn_pmc_group4.json', 8: 'attn_pmc_group8.json'}.get(clips_per_launch)
    if name is None:
        return None, None
    sha = attention_source_sha256()
    for rnd in sorted((d for d in os.listdir(os.path.join(ROOT, 'profiles')) if d.startswith('r')), reverse=True):
        try:
            with open(os.path.join(ROOT, 'profiles', rnd, name)) as f:
                j = json.load(f)
        except Exception:
            continue
        if j.get('attention_hip_sha256') == sha:
            return j['hbm_bytes_per_launch'], f'profiles/{rnd}/{name}'
        return None, f'profiles/{rnd}/{name} was measured on another attention.hip (stale)'
    return None, None


def self_launch(n: int) -> int:
    """--gpus N > 1 from a plain `python bench.py`: N fresh ranks as children of THIS process, which has not touched the GPU
    (never exec from a process that has).  stdout / stderr are inherited, so rank 0's JSON line is this command's output."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40, help='timed steps; a step = ONE FRAME OF EVERY CLIP IN FLIGHT on this GPU (see --step-unit)')
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--step-unit', default='batch', choices=['batch', 'frame'],
                    help="what one step propagates: 'batch' (default) = one frame of each of the --clips-in-flight clips, the batch this "
                         "engine advances together (24 frames); 'frame' = one frame of one clip (rounds 1-2: --steps 20 is then 5 group steps "
                         'from an idle GPU, a fill / drain transient rather than a rate)')
    ap.add_argument('--clips-in-flight', type=int, default=None,
                    help='clips advancing together on this GPU (default: RMEM_CLIPS_IN_FLIGHT, else the workload\'s own, else 24 = three groups of 8)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='davis17_480p_r50_N8', choices=sorted(WORKLOADS))
    ap.add_argument('--no-graphs', action='store_true')
    ap.add_argument('--clips-per-group', type=int, default=int(os.environ.get('RMEM_CLIPS_PER_GROUP', 8)),
                    help='> 1: that many clips advance in lockstep on one GroupEngine (one launch per layer for the group)')
    ap.add_argument('--host-frames', action='store_true',
                    help='PCIe-inclusive variant (not the contract line): frames start as decoded uint8 RGB in pinned host memory')
    ap.add_argument('--encoder-lookahead', type=int, default=int(os.environ.get('RMEM_ENC_LOOKAHEAD', 0)),
                    help='frames the ResNet-50 encoder runs ahead inside a clip (one launch per layer for all of them)')
    ap.add_argument('--feeder', default=None, choices=['queue', 'static'],
                    help='how the clip list reaches the ranks: job-wide ticket queue (default for N > 1) or static longest-first split')
    ap.add_argument('--dtype', default=None, choices=['bf16', 'fp16'],
                    help="16-bit operand type of the kernels (cfg.MODEL_DTYPE); default: the workload's (bf16; fp16 for the cfg-5 workload)")
    ap.add_argument('--roofline-launches', type=int, default=32, help='isolated T = 8 memory-read launches timed after the timed region')
    ap.add_argument('--drain', action='store_true',
                    help='run the WHOLE job once instead of a --steps window (non-cyclic clip list: a rank that finds the list empty '
                         'idles): value = propagated frames of all ranks / the slowest rank\'s seconds, tail and imbalance included')
    return ap.parse_args(argv)


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU execution path for the product')
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not os.environ.get('RMEM_SHARE_GPU'):
        raise SystemExit(f'rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible')
    local_rank %= ndev                      # RMEM_SHARE_GPU=1: rehearse the N > 1 code path on fewer GPUs (gloo backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist, store = None, None
    if world > 1:
        import torch.distributed as dist
        from rmem_ocu_amd.clip_runner import open_job_store
        store = open_job_store(rank, world)     # one explicit c10d store: the process group's and the clip queue's
        backend = os.environ.get('RMEM_DIST_BACKEND', 'nccl')       # nccl == RCCL over xGMI on ROCm
        if backend == 'nccl':
            dist.init_process_group('nccl', store=store, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, store=store, rank=rank, world_size=world)

    from rmem_ocu_amd import _lib, build_engine, build_vos_model, get_config
    from rmem_ocu_amd.clip_runner import DRAIN, ClipFeeder, ClipSlot, pump
    from rmem_ocu_amd.synth import make_clip, network_size
    from rmem_ocu_amd.weights import synth_state_dict

    wl = WORKLOADS[args.workload]
    VIDEO_HW, NUM_OBJS = wl['video'], wl['objs']
    cfg = get_config('pre_vost', 'bench', wl['model'])
    cfg.FORMER_MEM_LEN, cfg.LATTER_MEM_LEN = wl['former'], wl['latter']
    cfg.MODEL_DTYPE = args.dtype or wl.get('dtype', 'bf16')
    model = build_vos_model(cfg.MODEL_VOS, cfg).cuda(local_rank)
    deaot = cfg.MODEL_VOS == 'deaot'
    model.load_state_dict(synth_state_dict(0, encoder=cfg.MODEL_ENCODER, model='deaot' if deaot else 'aot'))
    net_hw = wl['net'] or network_size(*VIDEO_HW)

    if args.clips_in_flight is None:
        args.clips_in_flight = int(os.environ.get('RMEM_CLIPS_IN_FLIGHT', wl.get('in_flight', 24)))
    C = max(1, args.clips_in_flight)
    G = args.clips_per_group           # GroupEngine covers R50-AOTL, SwinB-AOTL and R50-DeAOTL
    # ---- the job: clips_per_gpu * world clips, lengths cycling through wl['lengths'] (whole groups per length) ----
    per_len = max(G, (wl['clips_per_gpu'] * world // len(wl['lengths'])) // G * G)
    lengths = [n for n in wl['lengths'] for _ in range(per_len)]
    feeder = ClipFeeder(lengths, rank, world, group=G, mode=args.feeder, store=store, cyclic=True)
    max_len = max(lengths)

    # clip CONTENT: two distinct synthetic clips per rank at the longest length (a clip id maps to one of them, truncated to
    # its own length): an 80-frame fp32 clip is 392 MB on the device, the clip list itself is only ids and lengths
    # (long clips: wl['distinct_frames'] generated frames played forwards and backwards up to the clip length)
    n_gen = min(max_len, wl.get('distinct_frames', max_len))
    clips_host = [make_clip(1000 * rank + j, n_gen, net_hw[0], net_hw[1], NUM_OBJS) for j in range(2)]
    if args.host_frames:
        # decoded video frames as a loader would hand them over: uint8 RGB [n, Hs, Ws, 3] at the VIDEO size in pinned memory;
        # every frame crosses PCIe (1.2 MB) and is resized + normalised on the device (rmem_ingest_rgb8)
        import torch.nn.functional as F
        clips = []
        for f, m in clips_host:
            v = F.interpolate(f, size=VIDEO_HW, mode='bilinear', align_corners=False)
            u8 = (v * 40.0 + 128.0).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().pin_memory()
            clips.append((u8, m.to(dev)))
    else:
        clips = [(f.to(dev), m.to(dev)) for f, m in clips_host]
    if n_gen < max_len:
        pingpong = (list(range(n_gen)) + list(range(n_gen - 2, 0, -1))) * (max_len // n_gen + 1)
        idx = torch.tensor(pingpong[:max_len])
        clips = [(f[idx] if f.is_cuda else f[idx].pin_memory(), m) for f, m in clips]

    def clip_data(cid):
        f, m = clips[cid % 2]
        return f[:lengths[cid]], m

    inject_at = wl.get('inject_at')
    new_obj = None
    if inject_at is not None:           # the new object's label map at the output size: one rectangle, label objs + 1
        if G <= 1 or args.host_frames:
            raise SystemExit(f'--workload {args.workload}: the new-object protocol runs on clip groups fed from HBM')
        new_obj = torch.zeros(VIDEO_HW[0], VIDEO_HW[1], dtype=torch.uint8, device=dev)
        new_obj[VIDEO_HW[0] // 2:VIDEO_HW[0] // 2 + VIDEO_HW[0] // 4, VIDEO_HW[1] // 8:VIDEO_HW[1] // 8 + VIDEO_HW[1] // 5] = NUM_OBJS + 1

    # frames the encoder runs ahead (0 = default: 2 with clip groups -- 8 images per launch --, 4 for single clips)
    lookahead = args.encoder_lookahead or (2 if G > 1 else 4)      # ResNet-50: encoder_batch.BatchEncoder, Swin-B: SwinBatchEncoder
    slots = []
    if G > 1:
        from rmem_ocu_amd.clip_runner import GroupSlot
        from rmem_ocu_amd.networks.engines.group_engine import GroupEngine
        C = max(1, C // G)                    # C groups of G clips each
        # stream creation order = hardware-queue assignment (round-robin over the runtime's 4 queues): 'pairs' (default) creates
        # main, encoder, main, encoder ...; 'mains_first' all main streams, then all encoder streams
        order = os.environ.get('RMEM_STREAM_ORDER', 'pairs')
        if order == 'mains_first':
            mains = [torch.cuda.Stream(dev) for _ in range(C)]
            encs = [torch.cuda.Stream(dev) for _ in range(C)]
            pools = list(zip(mains, encs))
        elif order == 'encs_first':
            encs = [torch.cuda.Stream(dev) for _ in range(C)]
            mains = [torch.cuda.Stream(dev) for _ in range(C)]
            pools = list(zip(mains, encs))
        else:
            pools = [None] * C
        for j in range(C):
            eng = GroupEngine(model, G, local_rank, 5, lookahead=lookahead, streams=pools[j])
            eng.use_graphs = not args.no_graphs
            slots.append(GroupSlot(eng, VIDEO_HW, dev))
        inner_of = lambda s: s.engine                                            # noqa: E731

        def start(s):
            ids = feeder.next_unit()
            if ids is None:
                return False                     # the job's list is drained (--drain): this slot idles
            data = [clip_data(i) for i in ids]
            s.start([d[0] for d in data], [d[1] for d in data], NUM_OBJS,
                    new_objects=None if new_obj is None else {c: (inject_at, new_obj) for c in range(len(ids))})
            return True
    else:
        for j in range(C):
            eng = build_engine(cfg.MODEL_ENGINE, phase='eval', aot_model=model, gpu_id=local_rank, long_term_mem_gap=5)
            eng.set_async(use_graphs=not args.no_graphs)
            slots.append(ClipSlot(eng, VIDEO_HW, dev, lookahead=lookahead))
        inner_of = lambda s: s.engine.aot_engines[0]                             # noqa: E731

        def start(s):
            ids = feeder.next_unit()
            if ids is None:
                return False
            s.start(*clip_data(ids[0]), NUM_OBJS)
            return True

    # ---- priming (untimed setup): every slot runs one whole clip, interleaved exactly like the timed region, which builds
    # every launch list / hipGraph (T = 1..8); then the slots are staggered so they sit at different clip positions ----
    for s in slots:
        start(s)
    while not all(s.done for s in slots):
        for s in slots:
            if not s.done:
                s.step()
    # frames one step propagates: one frame of every clip in flight (the batch the engine advances together), or one frame
    F = C * max(G, 1) if args.step_unit == 'batch' else 1
    if args.drain:
        # the whole job once, from empty slots: a fresh NON-cyclic list (its own ticket counter), every slot takes units until
        # the list is empty and then idles; the priming pass above was the warm-up
        feeder = ClipFeeder(lengths, rank, world, group=G, mode=args.feeder, store=store, cyclic=False)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        enc0 = sum(s.frames_encoded for s in slots)
        t0 = time.perf_counter()
        pump(slots, start, DRAIN, G)
        # propagated frames this rank ran (frame 0 of a clip is its reference frame): from the units it actually took
        frames_timed = sum(len(feeder.units[u]) * (lengths[feeder.units[u][0]] - 1) for u in feeder.history)
        args.steps = max(1, frames_timed // F)
        host_enqueue = time.perf_counter() - t0
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    else:
        for s in slots:
            start(s)
        n0 = min(lengths)
        for k in range(n0 - 1):
            for j, s in enumerate(slots):
                if k < (j * (n0 - 1)) // C and not s.done:
                    s.step()
        torch.cuda.synchronize()

        def run_steps(n):
            """n propagated frames in total (a group step propagates G frames); a finished slot takes the job's next unit."""
            pump(slots, start, n, G)

        run_steps(args.warmup * F)
        torch.cuda.synchronize()

        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        enc0 = sum(s.frames_encoded for s in slots)
        t0 = time.perf_counter()
        frames_timed = args.steps * F
        run_steps(frames_timed)
        host_enqueue = time.perf_counter() - t0
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    enc_in_window = sum(s.frames_encoded for s in slots) - enc0

    checksum = float(sum(int(s.labels.sum().item()) for s in slots))
    from rmem_ocu_amd.clip_runner import gather_stats
    gdev = dev if (dist is None or dist.get_backend() == 'nccl') else torch.device('cpu')
    agg = gather_stats(float(frames_timed), elapsed, checksum, dist, rank, world, gdev)   # the one data exchange: 24 bytes per rank

    # ---- roofline leg, after (and outside) the timed region: isolated launches of the dominant kernel under HIP events ----
    L = _lib.lib()
    roof = {'achieved': None, 'frac': None, 'launches_timed': 0, 'avg_launch_us': None, 't_mix': {}}
    by_T = {}
    if rank == 0 and args.roofline_launches > 0:
        inner = inner_of(slots[0])
        inner.stream.synchronize()
        torch.cuda.synchronize()
        s_int = inner.stream.cuda_stream
        ms, fl, nl = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        by_T = {}
        if deaot:       # the dominant kernel of this workload is the value-side GEMM of the long-term gated attention
            T = wl['former'] + wl['latter']
            op, _ = inner.rt.mem_read_probe(T)
            for _ in range(3):
                op(s_int)
            inner.stream.synchronize()
            _lib.check(L.rmem_gated_profile_start(), 'rmem_gated_profile_start')
            for _ in range(args.roofline_launches):
                op(s_int)
                inner.stream.synchronize()
            _lib.check(L.rmem_gated_profile_stop(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(nl)), 'rmem_gated_profile_stop')
        else:
            # bank sizes the workload actually reaches: N for a restricted bank; (8, 30) for the unbounded cfg-4 workload.  The LAST
            # one measured is the line's `roofline` (the steady state / the stress case), all of them are in roofline.by_T
            Ts = wl.get('roofline_T') or (wl['former'] + wl['latter'],)
            for T in Ts:
                op, _ = inner.rt.mem_read_probe(T)
                for _ in range(3):              # untimed: instruction cache, TLB
                    op(s_int)
                inner.stream.synchronize()
                _lib.check(L.rmem_profile_start(args.roofline_launches + 4), 'rmem_profile_start')
                for _ in range(args.roofline_launches):
                    op(s_int)
                    inner.stream.synchronize()      # alone on the GPU: the event bracket is the kernel's own duration
                _lib.check(L.rmem_profile_stop(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(nl)), 'rmem_profile_stop')
                if nl.value and ms.value > 0:
                    by_T[str(T)] = {'achieved': round(fl.value / (ms.value * 1e-3) / 1e12, 2), 'avg_launch_us': round(1e3 * ms.value / nl.value, 2),
                                    'launches_timed': nl.value}
        if nl.value and ms.value > 0:
            ach = (fl.value / (ms.value * 1e-3)) / 1e12
            roof = {'achieved': round(ach, 2), 'frac': round(ach / PEAK_BF16_TFLOPS, 4), 'launches_timed': nl.value,
                    'avg_launch_us': round(1e3 * ms.value / nl.value, 2), 't_mix': {str(T): nl.value}}

    if rank == 0:
        total_frames, elapsed, _ = agg
        traffic, traffic_src = pmc_traffic(G) if args.workload.startswith('davis17_480p_r50_N8') else (None, None)
        L16 = (net_hw[0] // 16 if wl['net'] else (net_hw[0] - 1) // 16 + 1) * (net_hw[1] // 16 if wl['net'] else (net_hw[1] - 1) // 16 + 1)
        out = {
            'metric': 'frames/sec (whole node) 480p VOS, N=8 memory bank' if args.workload == 'davis17_480p_r50_N8'
            else f'frames/sec (whole node) {args.workload}', 'value': round(total_frames / elapsed, 2), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / max(args.steps, 1), 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': cfg.MODEL_DTYPE, 'data': 'synthetic',
            'config': {'workload': args.workload, 'clip_frames': list(wl['lengths']) if len(wl['lengths']) > 1 else wl['lengths'][0],
                       'job_clips': len(lengths), 'clip_feeder': feeder.mode, 'video_hw': list(VIDEO_HW), 'network_hw': list(net_hw),
                       'tokens': L16, 'objects': NUM_OBJS,
                       'memory_bank': f"{wl['former']}+{wl['latter']}", 'gap': [max(int(round(n / 30)), 5) for n in wl['lengths']][0],
                       'clips_in_flight_per_gpu': C * G, 'clips_per_group': G, 'step_unit': args.step_unit, 'frames_per_step': F,
                       'frames_counted_in_timed_region': frames_timed, 'frames_executed_in_timed_region': -(-frames_timed // G) * G,
                       'frames_encoded_in_timed_region': enc_in_window,
                       'parallelism': f'clip-parallel x{world}', 'weights': 'synthetic (no checkpoint offline)',
                       'hipgraphs': not args.no_graphs, 'frames_from': 'pinned host uint8 (PCIe-inclusive)' if args.host_frames else 'HBM',
                       'encoder_lookahead': lookahead, 'host_enqueue_ms_per_frame': round(1e3 * host_enqueue / max(frames_timed, 1), 4),
                       'timed_region': 'whole job, drained (non-cyclic list)' if args.drain else 'window of --steps frames of a cyclic job',
                       **({'new_object_at_frame': inject_at} if inject_at is not None else {})},
            'roofline': {'bound': 'mfma', 'kernel': 'k_gp_pv<1, true>' if deaot else 'k_attn_partial<true, true>',
                         'achieved': roof['achieved'], 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': roof['frac'],
                         'traffic': traffic, 'traffic_source': traffic_src, 'launches_timed': roof['launches_timed'],
                         'avg_launch_us': roof['avg_launch_us'], 't_mix': roof['t_mix'],
                         'clips_per_launch': G, 'sampled': 'after the timed region, isolated direct launches',
                         **({'by_T': by_T} if len(by_T) > 1 else {})},
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == 'davis17_480p_r50_N8':
            out['cpu_baseline'], oracle_labels, oracle_indexes = cpu_baseline(*clips_host[0], VIDEO_HW)
            if G > 1 and not args.host_frames:
                out['check'] = check_against_oracle(slots[0], clips[0][0], clips[0][1], NUM_OBJS, oracle_labels, oracle_indexes)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
