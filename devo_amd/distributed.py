"""Multi-GPU harness helpers.  The update/BA hot path shards at SEQUENCE granularity: every
rank owns an independent sequence (its own pyramid, graph, poses, patches), so inference/benchmark runs are
replicas with NO data-path collective; the only communication is the timing barrier and a max-reduce of the
elapsed time.  (Training adds DDP's gradient all-reduce over RCCL — the reference's only collective,
train.py:107.)  Works with any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" on CPU."""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def barrier_sync(device=None):
    """barrier + device synchronize on both sides of a timed region (bench contract)."""
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    if world() > 1:
        dist.barrier()
    if device is not None and torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value, device=None):
    """MAX over ranks of a python float (the slowest rank defines the step time)."""
    if world() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_sequences(num_sequences, rank_=None, world_=None):
    """Indices of the independent sequences this rank owns (contiguous block partition, remainder to the
    first ranks) — mirrors DistributedSampler's role in train.py:91-93 without shuffling."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    base, rem = divmod(num_sequences, w)
    lo = r * base + min(r, rem)
    return list(range(lo, lo + base + (1 if r < rem else 0)))


def aggregate_throughput(units_per_rank, elapsed_s, device=None):
    """Whole-job throughput: total units over the max-over-ranks time."""
    total = units_per_rank
    if world() > 1:
        t = torch.tensor([float(units_per_rank)], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = float(t.item())
    return total / max_over_ranks(elapsed_s, device)


# ------------------------------------------------------------------------------------------------- process launcher
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _child(rank, nprocs, port, fn, args):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(nprocs), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the host driver only supports dmabuf IPC (RCCL needs it)
    fn(*args)


def launch(fn, nprocs, args=()):
    """One process per rank on THIS node (what `torch.distributed.run --nproc-per-node N` or train.py:253's mp.spawn do):
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in the environment, then fn(*args) in every child.
    `fn` must be importable (module-level).  Raises if any rank fails."""
    import torch.multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_child, args=(r, nprocs, port, fn, args)) for r in range(nprocs)]
    for p in procs:
        p.start()
    # poll: when one rank dies (an import error, an exception before or inside a collective) the others would wait in
    # RCCL / gloo forever — terminate them and raise, like torch.multiprocessing.spawn does
    import time
    bad, live = [], set(range(nprocs))
    while live and not bad:
        for r in sorted(live):
            p = procs[r]
            p.join(timeout=0.05)
            if p.exitcode is not None:
                live.discard(r)
                if p.exitcode != 0:
                    bad.append((r, p.exitcode))
        if live and not bad:
            time.sleep(0.05)
    if bad:
        for r in live:
            procs[r].terminate()
        for r in live:
            procs[r].join(timeout=10)
            if procs[r].is_alive():
                procs[r].kill()
        raise RuntimeError(f"launch: ranks failed (rank, exit code): {bad}" + (f"; terminated the remaining ranks {sorted(live)}" if live else ""))


def init_from_env(backend=None, device=None, force=False):
    """init_process_group from the launcher's environment; backend "nccl" (= RCCL) for GPU devices, "gloo" otherwise.
    A single process needs no group and gets none unless force=True (tests: the RCCL communicator path with world size 1).
    Returns (rank, world)."""
    import os
    w = int(os.environ.get("WORLD_SIZE", "1"))
    if (w > 1 or force) and not dist.is_initialized():
        if w == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        kw = {"device_id": torch.device(device)} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, **kw)
        if dist.get_world_size() != w:
            raise RuntimeError(f"world size {dist.get_world_size()} != WORLD_SIZE {w}")
    return rank(), world()
