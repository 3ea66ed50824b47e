"""world_size-2 gloo tests (CPU) of the multi-rank harness used by bench.py: the process launcher (devo_amd.distributed.launch,
what `bench.py --gpus N` spawns its ranks with), sequence sharding, barrier, max-over-ranks timing, whole-job aggregation, and
the data-parallel training wrapper — DistributedDataParallel over the 3 397 061-parameter bucket of devo_amd.training.TrainNet
(the reference's only collective, train.py:106-107)."""
import json
import os
import sys
import tempfile
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _harness_rank(outdir):
    sys.path.insert(0, ROOT)
    from devo_amd import distributed as D
    rank, world = D.init_from_env("gloo")
    mine = D.shard_sequences(5)
    D.barrier_sync(None)
    elapsed = 1.0 + rank                      # rank 1 is "slower"
    tmax = D.max_over_ranks(elapsed)
    agg = D.aggregate_throughput(len(mine) * 10, elapsed)
    with open(os.path.join(outdir, f"h{rank}.json"), "w") as f:
        json.dump([rank, mine, tmax, agg, D.world(), D.rank(), os.environ["LOCAL_RANK"], os.environ["MASTER_ADDR"]], f)
    dist.destroy_process_group()


def test_two_rank_harness_through_the_launcher():
    from devo_amd import distributed as D
    with tempfile.TemporaryDirectory() as td:
        D.launch(_harness_rank, 2, (td,))
        res = sorted(json.load(open(os.path.join(td, f"h{r}.json"))) for r in range(2))
    (r0, s0, t0, a0, w0, k0, l0, m0), (r1, s1, t1, a1, w1, k1, l1, m1) = res
    assert (w0, w1, k0, k1, l0, l1) == (2, 2, 0, 1, "0", "1") and m0 == m1 == "127.0.0.1"
    assert s0 == [0, 1, 2] and s1 == [3, 4]               # disjoint cover, remainder to the first rank
    assert t0 == t1 == 2.0                                 # MAX over ranks
    assert abs(a0 - 50 / 2.0) < 1e-12 and a0 == a1         # (30 + 20 units) / slowest rank


def _ddp_rank(outdir):
    sys.path.insert(0, ROOT)
    from devo_amd import distributed as D, training as T
    rank, world = D.init_from_env("gloo")
    net, model, opt = T.build_trainer("cpu", world)              # DDP(TrainNet) on CPU tensors: same wrapper, gloo instead of RCCL
    n = net.num_parameters()
    # a rank-dependent loss over EVERY parameter: d loss / d p = rank + 1, so the all-reduced (mean) gradient is 1.5 everywhere
    loss = model({"wiring_check": rank + 1})                     # through DDP's forward (that is what arms the gradient hooks)
    loss.backward()
    g = torch.cat([q.grad.reshape(-1) for q in net.parameters()])
    w0 = torch.cat([q.detach().reshape(-1) for q in net.parameters()]).double().sum().item()   # identical initial weights (broadcast)
    opt.step()
    with open(os.path.join(outdir, f"d{rank}.json"), "w") as f:
        json.dump([rank, n, int(g.numel()), float(g.min()), float(g.max()), w0], f)
    dist.destroy_process_group()


def test_ddp_allreduces_the_reference_sized_gradient_bucket():
    from devo_amd import distributed as D, training as T
    with tempfile.TemporaryDirectory() as td:
        D.launch(_ddp_rank, 2, (td,))
        res = sorted(json.load(open(os.path.join(td, f"d{r}.json"))) for r in range(2))
    for r, n, gn, gmin, gmax, w0 in res:
        assert n == gn == T.N_TOTAL == 3_397_061           # 13.59 MB of fp32 gradients, one bucket (train.py:107)
        assert gmin == gmax == 1.5                         # mean over the two ranks of (1, 2)
    assert res[0][5] == res[1][5]                          # both ranks hold the same parameters


def test_eight_rank_harness_and_ddp_bucket():
    """BASELINE configuration 4's process layout — 8 ranks of one node — on CPU over gloo: the launcher starts eight ranks with distinct
    LOCAL_RANKs on 127.0.0.1, the sequences are sharded disjointly, the timed region's MAX over ranks is the slowest rank's (8 s), the
    whole-job rate adds all ranks' units, and DDP's all-reduce of the 13.59 MB gradient bucket averages eight rank-dependent gradients
    (mean of 1 .. 8 = 4.5) into identical parameters everywhere.  RCCL with 8 ranks has run nowhere (one GPU per box): this is the same
    wrapper, launcher and reduction arithmetic with the other backend."""
    from devo_amd import distributed as D, training as T
    with tempfile.TemporaryDirectory() as td:
        D.launch(_harness_rank, 8, (td,))
        res = sorted(json.load(open(os.path.join(td, f"h{r}.json"))) for r in range(8))
    assert [r[0] for r in res] == list(range(8)) and all(r[4] == 8 and r[5] == r[0] and r[6] == str(r[0]) and r[7] == "127.0.0.1" for r in res)
    assert sorted(sum((r[1] for r in res), [])) == [0, 1, 2, 3, 4] and [len(r[1]) for r in res] == [1, 1, 1, 1, 1, 0, 0, 0]
    assert all(r[2] == 8.0 for r in res)                                   # MAX over ranks of 1 + rank
    assert all(abs(r[3] - 50 / 8.0) < 1e-12 for r in res)                  # 5 sequences x 10 units / the slowest rank
    with tempfile.TemporaryDirectory() as td:
        D.launch(_ddp_rank, 8, (td,))
        res = sorted(json.load(open(os.path.join(td, f"d{r}.json"))) for r in range(8))
    for r, n, gn, gmin, gmax, w0 in res:
        assert n == gn == T.N_TOTAL == 3_397_061
        assert gmin == gmax == 4.5
    assert len({r[5] for r in res}) == 1


def test_launcher_reports_a_failing_rank():
    from devo_amd import distributed as D
    import pytest
    with pytest.raises(RuntimeError, match="ranks failed"):
        D.launch(_failing_rank, 2, ())


def _failing_rank():
    if os.environ["RANK"] == "1":
        raise SystemExit(3)


def test_single_process_defaults():
    from devo_amd import distributed as D
    assert D.world() == 1 and D.rank() == 0
    assert D.shard_sequences(3) == [0, 1, 2]
    assert D.max_over_ranks(0.5) == 0.5
    assert D.aggregate_throughput(10, 2.0) == 5.0
