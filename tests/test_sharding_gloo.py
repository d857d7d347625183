"""The N > 1 path on CPU: ranks over gloo take a clip list from ClipFeeder (job-wide ticket queue on the TCPStore, or the
static longest-first split), 'run' their clips through clip_runner.pump with stub slots and rank 0 gathers
(frames, seconds, checksum) -- the only exchange the clip-parallel design has.  Also: bench.py --gpus N launches its own
ranks and propagates their exit code."""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubSlot:
    """Stands in for GroupSlot / ClipSlot: a unit of equal-length clips, one step = one frame of each."""

    def __init__(self, delay):
        self.done, self.left, self.ids, self.delay, self.ran = True, 0, [], delay, []

    def start(self, ids, n):
        self.ids, self.left, self.done = ids, n - 1, n <= 1
        self.ran.append(list(ids))

    def step(self):
        import time
        time.sleep(self.delay)
        self.left -= 1
        self.done = self.left <= 0


def _worker(rank, world, port, lengths, group, mode, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from rmem_ocu_amd.clip_runner import DRAIN, ClipFeeder, gather_stats, open_job_store, pump
    store = open_job_store(rank, world)                # the way bench.py does it: one explicit store for the group and the queue
    dist.init_process_group('gloo', store=store, rank=rank, world_size=world)
    first = ClipFeeder(lengths, rank, world, group=group, mode=mode, store=store)
    if mode == 'queue':                                # an earlier job of the same process group must not eat this job's tickets
        assert first.next_unit() is not None
    feeder = ClipFeeder(lengths, rank, world, group=group, mode=mode, store=store)
    slots = [_StubSlot(0.002 * (1 + 3 * rank)) for _ in range(2)]       # rank 1 is 4x slower
    frames = [0]

    def start(s):
        ids = feeder.next_unit()
        if ids is None:
            return False
        s.start(ids, lengths[ids[0]])
        frames[0] += sum(lengths[i] - 1 for i in ids)                   # propagated frames (frame 0 is the reference frame)
        return True

    import time
    dist.barrier()
    t0 = time.perf_counter()
    ran = pump(slots, start, DRAIN, group)              # bench.py --drain: the whole list once, idle when it is empty
    busy = time.perf_counter() - t0
    assert ran >= frames[0]                             # pump counts whole groups; a remainder unit holds fewer clips
    mine = [i for s in slots for u in s.ran for i in u]
    checksum = float(sum((i + 1) * lengths[i] for i in mine))
    dist.barrier()
    out = gather_stats(float(frames[0]), 1.0 + rank, checksum, dist, rank, world, torch.device('cpu'))
    counts = [None] * world
    dist.all_gather_object(counts, (len(mine), sorted(mine), busy))
    if rank == 0:
        q.put((out, counts))
    dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['queue', 'static'])
def test_two_rank_feeder_pump_and_gather(mode):
    # static: a skewed list (one clip is 40 % of the job); queue: many similar units, so the faster rank must end up with more
    _run_two_ranks(mode, [36, 80, 600, 12, 90, 300, 45, 45, 80, 80, 36, 36] if mode == 'static' else [40, 36] * 16)


def _drain_rate(mode):
    return _run_two_ranks(mode, [40, 36] * 16)


def _run_two_ranks(mode, lengths):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, 2, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    out, counts = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total_frames, max_seconds, checksum = out
    assert total_frames == sum(n - 1 for n in lengths)
    assert max_seconds == 2.0                                     # max over ranks, as bench.py reports
    assert checksum == sum((i + 1) * n for i, n in enumerate(lengths))
    all_ids = sorted(counts[0][1] + counts[1][1])
    assert all_ids == list(range(len(lengths)))                   # every clip ran exactly once, on exactly one rank
    if mode == 'queue':                                           # the fast rank took more of the list (work stealing)
        f0 = sum(lengths[i] for i in counts[0][1])
        f1 = sum(lengths[i] for i in counts[1][1])
        assert f0 > f1, (f0, f1)
    return total_frames / max(c[2] for c in counts)               # bench.py --drain's value: all frames / the slowest rank's seconds


def test_drained_job_shows_what_the_queue_buys():
    """The same list of similar clips drained by a fast and a 4x slower rank: with the static split both get half of the frames
    and the slow rank sets the time; with the ticket queue the fast rank takes most of the list.  Only a drain-the-job
    measurement (bench.py --drain) can show that -- a fixed window per rank cannot."""
    static = _drain_rate('static')
    queue = _drain_rate('queue')
    print(f'drained job: static {static:.0f} frames/s, queue {queue:.0f} frames/s')
    assert queue > 1.3 * static, (queue, static)


def test_group_units_and_static_feeder_single_rank():
    from rmem_ocu_amd.clip_runner import ClipFeeder, group_units
    lengths = [80, 40, 80, 80, 40, 80, 80]
    units = group_units(lengths, 4)
    assert units == [[0, 2, 3, 5], [6], [1, 4]]                   # equal lengths together, longest first, remainder units
    f = ClipFeeder(lengths, group=4)
    got = [f.next_unit() for _ in range(4)]
    assert got[:3] == [[0, 2, 3, 5], [6], [1, 4]] and got[3] is None      # most frames first, ties in list order
    c = ClipFeeder(lengths, group=4, cyclic=True)
    assert [c.next_unit() for _ in range(4)][3] == [0, 2, 3, 5]


def test_bench_self_launch_propagates_rank_failure():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (no GPU here: the ranks refuse to run, and the
    launcher must hand that failure on instead of printing a line)."""
    if torch.cuda.is_available():
        pytest.skip('needs a GPU-less host: on a GPU box this would run the benchmark')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'needs an MI355X' in (r.stdout + r.stderr)
    assert '"metric"' not in r.stdout
