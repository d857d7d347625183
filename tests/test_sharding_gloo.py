"""The N > 1 path on CPU: two ranks over gloo shard a clip list, 'process' their clips and rank 0
gathers (frames, seconds, checksum) -- the only exchange the clip-parallel design has."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lengths, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from rmem_ocu_amd.clip_runner import gather_stats, shard_clips
    mine = shard_clips(len(lengths), rank, world, lengths)
    frames = float(sum(lengths[i] - 1 for i in mine))            # propagated frames (frame 0 is the reference frame)
    seconds = 1.0 + rank                                          # rank 1 is the slow one
    checksum = float(sum((i + 1) * lengths[i] for i in mine))
    dist.barrier()
    out = gather_stats(frames, seconds, checksum, dist, rank, world, torch.device('cpu'))
    if rank == 0:
        q.put(out)
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    lengths = [36, 80, 600, 12, 90, 300, 45, 45, 80]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total_frames, max_seconds, checksum = out
    assert total_frames == sum(n - 1 for n in lengths)
    assert max_seconds == 2.0                                     # max over ranks, as bench.py reports
    assert checksum == sum((i + 1) * n for i, n in enumerate(lengths))
