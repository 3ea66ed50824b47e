"""bench.py's own N > 1 path, exercised before the driver does (one GPU shared by two ranks, `--share-gpu`): the launcher
(devo_amd.distributed.launch: one process per rank, RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1), the RCCL process group, the barrier +
max-over-ranks timing, the replica aggregation (`value` = N K / t), the data-parallel training probe (`train_dp`: DDP over the
reference's 3 397 061-parameter tree, the one collective of the path — train.py:31-42,90-95,106-107) and `--mode train`.
The JSON lines are kept under gpurun_out/ (copied to profiles/ by hand)."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("DEVO_BENCH_PROBE_TIMEOUT", "400")     # (two ranks on ONE GPU of a shared test box: the 90 s default is for a rank per GPU)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), lines[0]


_seen = {}


def _keep(name, line):
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name), "w") as f:
            f.write(line + "\n")


def test_update_op_bench_on_two_ranks_sharing_the_gpu():
    d, line = _run(["--gpus", "2", "--share-gpu", "--steps", "20", "--warmup", "5", "--no-f16"], timeout=900)
    _keep("bench_gpus2_share.json", line)
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak"
    assert "x2" in d["config"]["parallelism"] or "2" in d["config"]["parallelism"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert "train_dp" in d and "error" not in d["train_dp"], d.get("train_dp")
    assert d["train_dp"]["sequences_per_s"] > 0 and d["train_dp"]["parameters"] == 3_397_061 and d["train_dp"]["grad_bucket_bytes"] == 13_588_244
    _seen["train_dp_ms"] = d["train_dp"]["ms_per_step"]


def test_eight_ranks_first_call_work_stays_inside_the_probe_bound():
    """What the driver's first real 8-GPU run will do on every rank at once — load the library, build the inputs, hit MIOpen's find DB
    (devo_amd/miopen_db/) for the training probe's convolutions — here with all eight ranks on ONE GPU and the probe's DEFAULT bound of
    90 s (no DEVO_BENCH_PROBE_TIMEOUT): the line carries `train_dp` without an error, i.e. eight ranks' cold-start work fits the bound
    even when they share a device (52 s for the whole command on an MI355X box of this pool)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("DEVO_BENCH_PROBE_TIMEOUT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--steps", "18", "--warmup", "2", "--no-f16", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    _keep("bench_gpus8_share.json", lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 18 and d["scaling"] == "weak" and d["value"] > 0
    assert "train_dp" in d and "error" not in d["train_dp"], d.get("train_dp")
    assert d["train_dp"]["parameters"] == 3_397_061 and d["train_dp"]["grad_bucket_bytes"] == 13_588_244


def test_training_bench_on_two_ranks_sharing_the_gpu():
    d, line = _run(["--mode", "train", "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "4", "--train-iters", "2"], timeout=900)
    _keep("bench_train_gpus2_share.json", line)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "dp2" in d["config"]["parallelism"] or "2" in d["config"]["parallelism"]
    # (the probe inside the update-op run measures the same step; its figure is NOT compared here: with two ranks time-slicing one GPU of a
    #  fresh box, MIOpen's first-call searches of both ranks land in either measurement — 5 s against 0.1 s has been seen)
    assert _seen.get("train_dp_ms", 1.0) > 0


def test_rccl_process_group_with_one_rank():
    """`nccl` (= RCCL on ROCm) has to load, create a communicator next to live HIP work and run a collective on an MI355X before the
    driver's 8-GPU run depends on it: one rank, init_process_group("nccl", device_id=...), an all-reduce, DistributedDataParallel's
    bucketed gradient all-reduce of a small module, destroy."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "from devo_amd import distributed as D\n"
        "dev = torch.device('cuda', 0); torch.cuda.set_device(dev)\n"
        "x = torch.ones(1 << 20, device=dev) * 3\n"
        "g = torch.cuda.CUDAGraph()\n"
        "y = torch.zeros_like(x)\n"
        "s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())\n"
        "with torch.cuda.stream(s):\n"
        "    with torch.cuda.graph(g, stream=s):\n"
        "        y.copy_(x * 2)\n"
        "torch.cuda.current_stream().wait_stream(s); g.replay()\n"
        "r, w = D.init_from_env('nccl', dev, force=True)\n"
        "assert (r, w) == (0, 1) and dist.get_backend() == 'nccl'\n"
        "dist.all_reduce(x); dist.barrier(); torch.cuda.synchronize()\n"
        "assert float(x[0]) == 3.0 and float(y[0]) == 6.0\n"
        "m = torch.nn.parallel.DistributedDataParallel(torch.nn.Linear(256, 256).to(dev), device_ids=[0])\n"
        "m(torch.randn(8, 256, device=dev)).square().mean().backward(); torch.cuda.synchronize()\n"
        "assert m.module.weight.grad is not None and bool(torch.isfinite(m.module.weight.grad).all())\n"
        "g.replay(); torch.cuda.synchronize()\n"
        "dist.destroy_process_group(); print('rccl ok')\n")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("MASTER_PORT", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "rccl ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


def test_update_op_bench_under_the_drivers_launch_command():
    """the contract's launch for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` — bench.py reads RANK / LOCAL_RANK / WORLD_SIZE from the environment instead of spawning;
    rank 0 prints the one JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "20", "--warmup", "5", "--no-f16", "--no-cpu-baseline", "--no-train-probe"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    _keep("bench_torchrun_gpus2_share.json", lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0 and d["scaling"] == "weak"


def test_training_bench_through_rccl_under_the_drivers_launch_command():
    """BASELINE configuration 4's code path on the real backend, as far as one GPU allows: the driver's launch command with one rank,
    `--mode train --ddp-single` = init_process_group("nccl") + DistributedDataParallel + the gradient all-reduce over RCCL (one rank); the
    line names the parallelism and the collective the way the N = 8 line will."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29543",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "train", "--ddp-single", "--steps", "2", "--warmup", "2", "--train-iters", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    _keep("bench_train_rccl_1rank.json", lines[0])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1" and d["scaling"] == "weak" and d["value"] > 0
    assert d["train"]["collective"] == "DDP all-reduce (RCCL)" and d["train"]["grad_bucket_bytes"] == 13_588_244


def test_multi_gpu_update_op_line_skips_the_single_gpu_probes():
    """An N > 1 line must come out quickly on the driver's scaling run: no reference-API probe, no full-iteration probe, no CPU baseline
    (rank 0 would hold the other ranks in the final barrier), the training probe bounded by DEVO_BENCH_PROBE_TIMEOUT (90 s by default)."""
    import time
    t0 = time.time()
    d, _ = _run(["--gpus", "2", "--share-gpu", "--steps", "20", "--warmup", "5"], timeout=600)
    assert "reference_api" not in d and "full_update_iteration" not in d and "cpu_baseline" not in d
    assert "f16" in d and "train_dp" in d
    assert time.time() - t0 < 300
