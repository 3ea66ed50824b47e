"""world_size-2 gloo test (CPU) of the multi-rank harness used by bench.py: sequence sharding, barrier,
max-over-ranks timing and whole-job aggregation."""
import os
import sys
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from devo_amd import distributed as D
    mine = D.shard_sequences(5)
    D.barrier_sync(None)
    elapsed = 1.0 + rank                      # rank 1 is "slower"
    tmax = D.max_over_ranks(elapsed)
    agg = D.aggregate_throughput(len(mine) * 10, elapsed)
    q.put((rank, mine, tmax, agg, D.world(), D.rank()))
    dist.destroy_process_group()


def test_two_rank_harness():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, t0, a0, w0, k0), (r1, s1, t1, a1, w1, k1) = res
    assert (w0, w1, k0, k1) == (2, 2, 0, 1)
    assert s0 == [0, 1, 2] and s1 == [3, 4]               # disjoint cover, remainder to the first rank
    assert t0 == t1 == 2.0                                 # MAX over ranks
    assert abs(a0 - 50 / 2.0) < 1e-12 and a0 == a1         # (30 + 20 units) / slowest rank


def test_single_process_defaults():
    sys.path.insert(0, ROOT)
    from devo_amd import distributed as D
    assert D.world() == 1 and D.rank() == 0
    assert D.shard_sequences(3) == [0, 1, 2]
    assert D.max_over_ranks(0.5) == 0.5
    assert D.aggregate_throughput(10, 2.0) == 5.0
