"""Round 6 of fastba: the retraction on the solver's launch (k_ba_solve_retract) and cuda_ba.forward's prepared-table cache
(the index half of ba_cuda.cu:435-437 depends on kk alone; DEVO runs the BA many times on one patch graph).  Both change how the
work is launched, never a result: everything here is an equality of bits."""
import os
import subprocess
import sys
import pytest
import torch
from devo_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scene(nk=12, M=48, seed=9, shuffle=False):
    poses = synth.make_poses(nk, seed).to(DEV)
    patches = synth.make_patches(nk, M, 120, 160, seed=seed)[0].to(DEV)
    intr = synth.make_intrinsics(nk, 120, 160).to(DEV)
    ii, jj, kk = synth.full_graph(nk, M)
    if shuffle:
        p = torch.randperm(len(ii), generator=torch.Generator().manual_seed(seed))[: int(0.9 * len(ii))]
        ii, jj, kk = ii[p], jj[p], kk[p]
    ii, jj, kk = [t.to(DEV) for t in (ii, jj, kk)]
    delta, weight = [t.to(DEV) for t in synth.make_update_outputs(len(ii), seed, sigma=0.3)]
    from devo_amd.backends import cuda_ba
    c = cuda_ba.transform(poses, patches, intr, ii, jj, kk, layout="2pp")
    return poses, patches, intr, c[:, :, :, 1, 1] + delta, weight, torch.tensor([1e-4], device=DEV), ii, jj, kk


_FUSE_AB = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from devo_amd import synth
from devo_amd.backends import cuda_ba
out = {}
for nk, M, t0, shuffle in ((2, 40, 1, False), (3, 7, 1, False), (8, 300, 1, True), (12, 48, 1, False), (15, 96, 1, False), (15, 96, 5, True), (16, 80, 0, True), (20, 30, 1, False),
                           (21, 64, 0, True)):
    poses = synth.make_poses(nk, nk).cuda(); patches = synth.make_patches(nk, M, 120, 160, seed=nk)[0].cuda()
    intr = synth.make_intrinsics(nk, 120, 160).cuda()
    ii, jj, kk = synth.full_graph(nk, M)
    if shuffle:
        p = torch.randperm(len(ii), generator=torch.Generator().manual_seed(nk))[: int(0.9 * len(ii))]
        ii, jj, kk = ii[p], jj[p], kk[p]
    ii, jj, kk = [t.cuda() for t in (ii, jj, kk)]
    delta, weight = [t.cuda() for t in synth.make_update_outputs(len(ii), nk, sigma=0.3)]
    c = cuda_ba.transform(poses, patches, intr, ii, jj, kk, layout="2pp")
    cuda_ba.forward(poses, patches, intr, c[:, :, :, 1, 1] + delta, weight, torch.tensor([1e-4]).cuda(), ii, jj, kk, t0, nk, 2)
    out[f"{nk}_{M}_{t0}"] = (poses.cpu(), patches.cpu(), cuda_ba.last_path())
torch.save(out, sys.argv[2])
"""


def test_the_retraction_on_the_solvers_launch_changes_no_bit(tmp_path):
    """k_ba_solve_retract (G workgroups factorise the same system, each retracts its own patches from the solution in its LDS) against the
    two launches of rounds 1-5 (DEVO_BA_FUSE_RETRACT=0; read once per process: sub-processes) — windows of 1 .. 21 optimised poses, more
    patches than one workgroup's waves, shuffled edge lists, t0 = 0 / 1 / 5.  Up to 16 optimised poses every summation order is fixed:
    the same bits; beyond, the general accumulate kernel's atomics move the last bits of the system itself."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for fuse in (True, False):
        env = dict(os.environ)
        env.pop("DEVO_BA_FUSE_RETRACT", None)
        if not fuse:
            env["DEVO_BA_FUSE_RETRACT"] = "0"
        path = str(tmp_path / f"fuse_{int(fuse)}.pt")
        r = subprocess.run([sys.executable, "-c", _FUSE_AB, root, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res.append(torch.load(path))
    assert res[0].keys() == res[1].keys()
    for k in res[0]:
        nk, _, t0 = (int(x) for x in k.split("_"))
        assert torch.isfinite(res[0][k][0]).all() and torch.isfinite(res[0][k][1]).all()
        if nk - t0 <= 16:
            assert torch.equal(res[0][k][0], res[1][k][0]) and torch.equal(res[0][k][1], res[1][k][1]), k
        else:
            assert torch.allclose(res[0][k][0], res[1][k][0], atol=1e-4) and torch.allclose(res[0][k][1], res[1][k][1], atol=1e-4), k


def test_forward_remembers_the_index_tables_of_an_unchanged_kk():
    """cuda_ba.forward without a workspace argument (the reference's call, ba.cpp:153): the second call on the same kk tensor takes the
    prepared tables of the first (one launch less), an in-place edit of kk (version counter) or another tensor rebuilds them, and the
    results are the bits of a call that prepares."""
    from devo_amd.backends import cuda_ba
    poses, patches, intr, target, weight, lm, ii, jj, kk = _scene(shuffle=True)
    cuda_ba.prep_invalidate()
    h0, m0 = cuda_ba.prep_stats()

    def run(kk_):
        P, Q = poses.clone(), patches.clone()
        cuda_ba.forward(P, Q, intr, target, weight, lm, ii, jj, kk_, 1, 12, 2)
        return P, Q

    a = run(kk)
    assert cuda_ba.prep_stats() == (h0, m0 + 1)
    b = run(kk)
    c = run(kk)
    assert cuda_ba.prep_stats() == (h0 + 2, m0 + 1)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    # the same values in another tensor: a miss (nothing is assumed about storage the cache does not hold)
    d = run(kk.clone())
    assert cuda_ba.prep_stats() == (h0 + 2, m0 + 2) and torch.equal(a[0], d[0]) and torch.equal(a[1], d[1])
    # an in-place edit bumps the version counter: the tables are rebuilt for the NEW contents
    kk2 = kk.clone()
    ref_before = run(kk2)
    h1, m1 = cuda_ba.prep_stats()
    # two patches of one source frame trade their edges: another graph, still regular (one source frame per patch, distinct target frames:
    # fixed summation orders, comparable bit by bit)
    a_, b_ = 48 * 3 + 5, 48 * 3 + 17
    kk2.copy_(torch.where(kk2 == a_, torch.full_like(kk2, b_), torch.where(kk2 == b_, torch.full_like(kk2, a_), kk2)))
    got = run(kk2)
    assert cuda_ba.prep_stats() == (h1, m1 + 1)
    P, Q = poses.clone(), patches.clone()
    ws = cuda_ba.workspace(len(ii), patches.shape[1], 11, DEV)
    cuda_ba.prepare(kk2, patches.shape[1], 11, ws)
    cuda_ba.forward(P, Q, intr, target, weight, lm, ii, jj, kk2, 1, 12, 2, ws=ws, prepared=True)
    assert torch.equal(got[0], P) and torch.equal(got[1], Q) and not torch.equal(got[0], ref_before[0])


def test_an_explicit_prepare_or_a_capture_never_leaves_stale_tables_behind():
    """prepare() on any workspace drops what forward() remembers; a call captured into a HIP graph neither consults nor fills the cache
    (a capture executes nothing) and the replayed graph prepares by itself."""
    from devo_amd.backends import cuda_ba
    poses, patches, intr, target, weight, lm, ii, jj, kk = _scene(nk=10, M=40, seed=4)
    P0, Q0 = poses.clone(), patches.clone()
    cuda_ba.forward(P0, Q0, intr, target, weight, lm, ii, jj, kk, 1, 10, 2)
    h, m = cuda_ba.prep_stats()
    ws = cuda_ba.workspace(len(ii), patches.shape[1], 9, DEV)
    cuda_ba.prepare(kk, patches.shape[1], 9, ws)
    P1, Q1 = poses.clone(), patches.clone()
    cuda_ba.forward(P1, Q1, intr, target, weight, lm, ii, jj, kk, 1, 10, 2)
    assert cuda_ba.prep_stats() == (h, m + 1) and torch.equal(P0, P1) and torch.equal(Q0, Q1)
    # capture
    Pg, Qg = poses.clone(), patches.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            cuda_ba.forward(Pg, Qg, intr, target, weight, lm, ii, jj, kk, 1, 10, 2)
    torch.cuda.current_stream().wait_stream(s)
    h2, m2 = cuda_ba.prep_stats()
    assert (h2, m2) == (h, m + 1)                             # neither a hit nor a miss was counted during the capture
    Pg.copy_(poses); Qg.copy_(patches)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(Pg, P0) and torch.equal(Qg, Q0)
    P2, Q2 = poses.clone(), patches.clone()
    cuda_ba.forward(P2, Q2, intr, target, weight, lm, ii, jj, kk, 1, 10, 2)      # after the capture: a miss (the cache was dropped), same bits
    assert cuda_ba.prep_stats() == (h, m + 2) and torch.equal(P2, P0) and torch.equal(Q2, Q0)


def test_a_ba_call_is_at_most_seven_launches():
    """north_star / verdict r05 item 1: accumulate + reduce + (solve | retract) per Gauss-Newton iteration, the index tables once per graph."""
    from torch.profiler import profile, ProfilerActivity
    from devo_amd.backends import cuda_ba
    poses, patches, intr, target, weight, lm, ii, jj, kk = _scene(nk=15, M=96, seed=2)
    P, Q = poses.clone(), patches.clone()
    cuda_ba.forward(P, Q, intr, target, weight, lm, ii, jj, kk, 1, 15, 2)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        cuda_ba.forward(P, Q, intr, target, weight, lm, ii, jj, kk, 1, 15, 2)
        torch.cuda.synchronize()
    names = [ev.name for ev in prof.events() if "cuda" in str(getattr(ev, "device_type", "")).lower() and "k_ba" in ev.name]
    assert len(names) == 6 and sum("solve_retract" in n for n in names) == 2 and not any("k_ba_retract" in n or "prepare" in n for n in names), names


def test_index_tables_of_the_sliding_window_graph_with_devos_buffer_size():
    """cuda_ba.prepare on DEVO's steady-state edge list (45 312 edges in devo.py's order: the multi-kernel path, which works on the RANGE of patch ids)
    with the patch slots of DEVO's 2048-frame buffers (196 608): sorted unique ids and per patch exactly its edges in ascending order, as torch.unique
    groups them (ba_cuda.cu:435-437)."""
    from devo_amd.backends import cuda_ba
    kk = synth.sliding_window_graph(40, 96)[2]
    E, Np = len(kk), 2048 * 96
    ws = cuda_ba.workspace(E, Np, 10, torch.device(DEV))
    cuda_ba.prepare(kk.to(DEV), Np, 10, ws)
    n_seg, kx, seg, perm = cuda_ba.prepared_tables(ws, E, Np, 10)
    kx_ref, inv = torch.unique(kk, sorted=True, return_inverse=True)
    assert n_seg == len(kx_ref) == 22 * 96 and torch.equal(kx.cpu().long(), kx_ref)
    seg, perm = seg.cpu().long(), perm.cpu().long()
    assert seg[0] == 0 and seg[-1] == E and torch.equal(seg[1:] - seg[:-1], torch.bincount(inv, minlength=n_seg))
    assert torch.equal(inv[perm], torch.repeat_interleave(torch.arange(n_seg), seg[1:] - seg[:-1]))
    same = inv[perm][1:] == inv[perm][:-1]
    assert bool((perm[1:][same] > perm[:-1][same]).all())
    # the full table (sync=False): entries beyond n_seg read E — any reader sees empty tails
    n_dev, _, seg_full, _ = cuda_ba.prepared_tables(ws, E, Np, 10, sync=False)
    assert int(n_dev) == n_seg and bool((seg_full[n_seg:] == E).all())


_PREP_AB = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from devo_amd.backends import cuda_ba
out = {}
g = torch.Generator().manual_seed(7)
for name, E, Np in (("random", 20000, 900), ("bad_ids", 9000, 400), ("all_bad", 3000, 50), ("ascending", 21600, 1440)):
    kk = torch.randint(0, Np, (E,), generator=g)
    if name == "bad_ids":
        kk[::13] = -1; kk[5::17] = Np + 3
    if name == "all_bad":
        kk[:] = Np + 1
    if name == "ascending":
        kk = torch.arange(Np).repeat_interleave(E // Np)
    ws = cuda_ba.workspace(E, Np, 5, "cuda")
    cuda_ba.prepare(kk.cuda(), Np, 5, ws)
    n, kx, seg, perm = cuda_ba.prepared_tables(ws, E, Np, 5)
    out[name] = (n, kx.cpu(), seg.cpu(), perm.cpu())
torch.save(out, sys.argv[2])
"""


def test_the_two_preparation_paths_build_the_same_tables(tmp_path):
    """The single-workgroup kernel (<= 32 768 edges) and the multi-kernel path (beyond; DEVO_BA_PREP_MULTI_FROM forces it here) on the same lists —
    random ids, ids out of range (they join segment 0, as before), nothing but bad ids, ascending ids: identical tables."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for multi in (False, True):
        env = dict(os.environ)
        env.pop("DEVO_BA_PREP_MULTI_FROM", None)
        if multi:
            env["DEVO_BA_PREP_MULTI_FROM"] = "1"
        path = str(tmp_path / f"prep_{int(multi)}.pt")
        r = subprocess.run([sys.executable, "-c", _PREP_AB, root, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res.append(torch.load(path))
    for name in res[0]:
        a, b = res[0][name], res[1][name]
        assert a[0] == b[0], name
        for x, y in zip(a[1:], b[1:]):
            assert torch.equal(x, y), name


@pytest.mark.parametrize("nk,M,t0,iters,shuffle", [(12, 48, 1, 2, False), (15, 96, 1, 2, False), (15, 96, 5, 1, True), (9, 40, 9, 2, False), (24, 30, 1, 2, False),
                                                  (15, 96, 1, 0, False)])
def test_the_next_lookups_plan_rides_on_the_solvers_launch(nk, M, t0, iters, shuffle):
    """devo_ba_forward_prepared_delta_plan: the ordering step of a locality plan carried by extra workgroups of the first Gauss-Newton
    iteration's solver launch (k_ba_solve_retract_order).  The BA's results are the bits of forward_delta without the rider; the plan is the
    one cuda_corr.plan_finish makes from the same bins (a permutation with the heavy list in front and the same (frame, band) key sequence
    behind it; the same counts); the lookup under it returns the bits of the lookup under plan_finish's plan.  Cases without a fused solver
    launch (structure-only t0 = t1, 23 optimised poses, zero iterations) order the plan with a launch of their own."""
    from devo_amd.backends import cuda_ba, cuda_corr
    H, W, R = 120, 160, 3
    poses = synth.make_poses(nk, nk).to(DEV)
    patches = synth.make_patches(nk, M, H, W, seed=nk)[0].to(DEV)
    intr = synth.make_intrinsics(nk, H, W).to(DEV)
    ii, jj, kk = synth.full_graph(nk, M)
    if shuffle:
        p = torch.randperm(len(ii), generator=torch.Generator().manual_seed(nk))[: int(0.9 * len(ii))]
        ii, jj, kk = ii[p], jj[p], kk[p]
    ii, jj, kk = [t.to(DEV) for t in (ii, jj, kk)]
    E = len(ii)
    delta, weight = [t.to(DEV) for t in synth.make_update_outputs(E, nk, sigma=0.3)]
    lm = torch.tensor([1e-4], device=DEV)
    Np = patches.shape[1]
    outs, plans, coords = [], [], None
    for ride in (False, True):
        P_, Q_ = poses.clone(), patches.clone()
        ws = cuda_ba.workspace(E, Np, nk - t0, torch.device(DEV))
        cuda_ba.prepare(kk, Np, nk - t0, ws)
        c, buf = cuda_ba.transform(P_, Q_, intr, ii, jj, kk, layout="2pp", plan_for=(nk, H, R, W, 0))
        coords = c
        if ride:
            cuda_ba.forward_delta(P_, Q_, intr, c, delta, weight, lm, ii, jj, kk, t0, nk, iters, ws, plan_next=(buf, nk, H, W, 0))
        else:
            cuda_corr.plan_finish(buf, jj, nk, H, R, width=W)
            cuda_ba.forward_delta(P_, Q_, intr, c, delta, weight, lm, ii, jj, kk, t0, nk, iters, ws)
        outs.append((P_.cpu(), Q_.cpu()))
        plans.append(buf)
    if nk - t0 <= 16:
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    else:                                                        # (beyond 16 optimised poses the accumulation uses float atomics: no two calls agree to the bit)
        assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-4, atol=1e-5) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-4, atol=1e-5)
    a, b = plans[0].cpu(), plans[1].cpu()
    nh = int(a[E])
    assert nh == int(b[E]) and int(a[2 * E + 1]) == int(b[2 * E + 1])                       # heavy and dead counts
    assert sorted(b[:E].tolist()) == list(range(E)) and sorted(a[:nh].tolist()) == sorted(b[:nh].tolist())
    bins = a[E + 1:2 * E + 1]                                                                # (the scratch half keeps the bins)
    assert torch.equal(bins, b[E + 1:2 * E + 1])
    assert torch.equal(bins[a[nh:E].long()], bins[b[nh:E].long()])                           # the same bin at every slot
    # the lookup under either plan: the same bits (the order never enters a result)
    g = torch.Generator().manual_seed(3)
    fmap = (torch.randn(1, nk, 128, H, W, generator=g) * 0.5).to(DEV)
    gmap = (torch.randn(1, Np, 128, 3, 3, generator=g) * 0.5).to(DEV)
    Dm = 2 * R + 1
    res = []
    for buf in plans:
        out = torch.zeros(1, E, Dm * Dm * 9, device=DEV)
        cuda_corr.forward_into(out, gmap, fmap, coords, kk, jj, R, Dm * Dm * 9, 1, 0, order=buf)
        res.append(out)
    assert torch.equal(res[0], res[1]) and float(res[0].abs().max()) > 0
