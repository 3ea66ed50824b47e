"""GPU parity of fastba (cuda_ba.*) against the CPU oracle (oracle/fastba.py, oracle/pops.py):
bundle adjustment (in place, 1-2 iterations, fixed-pose window, structure-only, shuffled / duplicated edges,
patches whose edges have different source frames), unique(kk) compaction, neighbors (bit-exact), reproject,
and the fused transform (vs the oracle restatement of projective_ops.transform and vs the reference goldens).
Tolerance: integer outputs bit-exact; fp32 poses / inverse depths within 1e-4 (north_star) of the fp64 oracle."""
import os
import numpy as np
import pytest
import torch
from oracle import fastba as F
from oracle import pops
from oracle.lie import SE3 as OSE3
from devo_amd import synth
from util import assert_rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def scene(n=8, M=12, H=60, W=80, seed=3, sigma=1.0, keep=0.85, shuffle=True):
    poses = synth.make_poses(n, seed)
    patches, _ = synth.make_patches(n, M, H, W, seed=seed)
    intr = synth.make_intrinsics(n, H, W)
    ii, jj, kk = synth.full_graph(n, M)
    g = torch.Generator().manual_seed(seed)
    if shuffle:
        perm = torch.randperm(len(ii), generator=g)[: int(keep * len(ii))]
        ii, jj, kk = ii[perm], jj[perm], kk[perm]
    delta, weight = synth.make_update_outputs(len(ii), seed, sigma=sigma)
    c0 = pops.transform(OSE3(poses.double()), patches.double(), intr.double(), ii, jj, kk)
    target = (c0[..., 1, 1, :] + delta.double()).float()
    return poses, patches, intr, target, weight, ii, jj, kk


def run_ba(poses, patches, intr, target, weight, ii, jj, kk, t0, t1, iters):
    from devo_amd import fastba
    from devo_amd.lietorch import SE3
    P = poses.clone().to(DEV)
    Q = patches.clone().to(DEV)
    lm = torch.as_tensor([1e-4], device=DEV)
    out = fastba.BA(SE3(P), Q, intr.to(DEV), target.to(DEV), weight.to(DEV), lm, ii.to(DEV), jj.to(DEV), kk.to(DEV), t0, t1, iters)
    assert out == []
    return P.cpu(), Q.cpu()


def check(got, ref64, tol=1e-4, ref32=None):
    """per ROW (tests/util.py:row_rel_err): every pose's translation and quaternion and every patch's inverse depth within `tol` of
    ITSELF (down to 1 % of the tensor's scale) — not max-abs over the tensor's maximum, which lets a small component hide.
    ref32 = the oracle run in fp32 on the same inputs: where fp32 arithmetic itself moves a row by more than tol / 2 (an ill-conditioned
    patch), the bound for that row is twice the oracle's own fp32 error (the envelope of test_gpu_fullsize's 1-px test)."""
    from util import row_rel_err
    p, q = got
    assert torch.isfinite(p).all() and torch.isfinite(q).all()
    rows = lambda P, Q: (("translation", P[0, :, :3]), ("quaternion", P[0, :, 3:]), ("inverse depth", Q[0, :, 2].reshape(-1)))
    for i, (name, a) in enumerate(rows(p, q)):
        b = rows(*ref64)[i][1]
        err = row_rel_err(a, b)
        bound = torch.full_like(err, tol)
        if ref32 is not None:
            bound = torch.maximum(bound, 2.0 * row_rel_err(rows(*ref32)[i][1].double(), b))
        assert bool((err <= bound).all()), f"{name}: per-row relative error {float(err.max()):.3e}, worst error / bound {float((err / bound).max()):.2f} (tol {tol:.1e})"
    assert torch.equal(q[:, :, :2], ref64[1][:, :, :2].float())


@pytest.mark.parametrize("iters", [1, 2])
@pytest.mark.parametrize("t0", [1, 3])
def test_ba_matches_oracle(iters, t0):
    s = scene()
    n = s[0].shape[1]
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], t0, n, iters, dtype=torch.float64)
    check(run_ba(*s, t0, n, iters), ref)


def test_ba_structure_only_and_unobserved_patches():
    s = scene(keep=0.5)
    n = s[0].shape[1]
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], n, n, 2, dtype=torch.float64)
    got = run_ba(*s, n, n, 2)
    assert torch.equal(got[0], s[0])
    check(got, ref)


def test_ba_cfg_sizes_and_duplicates():
    """closer to cfg2: n=15 (14 optimised poses), M=20; plus duplicated edges and an unsorted list"""
    s = list(scene(n=15, M=20, H=120, W=160, seed=5, keep=1.0))
    dup = torch.arange(0, len(s[5]), 7)
    for k in (3, 4):
        s[k] = torch.cat([s[k], s[k][:, dup]], 1)
    for k in (5, 6, 7):
        s[k] = torch.cat([s[k], s[k][dup]])
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], 1, 15, 2, dtype=torch.float64)
    check(run_ba(*s, 1, 15, 2), ref)


def test_ba_patch_with_mixed_source_frames():
    """ii is taken per edge (ba_cuda.cu:241), not per patch: edges of one patch may name different source frames."""
    s = list(scene(seed=9))
    g = torch.Generator().manual_seed(1)
    s[5] = torch.where(torch.rand(len(s[5]), generator=g) < 0.3, torch.randint(0, 8, (len(s[5]),), generator=g), s[5])
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], 1, 8, 1, dtype=torch.float64)
    ref32 = F.ba(*s[:5], torch.tensor([1e-4]), *s[5:], 1, 8, 1, dtype=torch.float32)
    check(run_ba(*s, 1, 8, 1), ref, tol=2e-4, ref32=ref32)


def test_ba_requires_contiguous_inplace_buffers():
    from devo_amd.backends import cuda_ba
    s = scene()
    bad = s[0].to(DEV).repeat(1, 1, 2)[..., ::2]
    with pytest.raises(RuntimeError):
        cuda_ba.forward(bad, s[1].to(DEV), s[2].to(DEV), s[3].to(DEV), s[4].to(DEV), torch.tensor([1e-4], device=DEV),
                        s[5].to(DEV), s[6].to(DEV), s[7].to(DEV), 1, 8, 2)


def test_neighbors_bit_exact():
    from devo_amd import fastba
    g = torch.Generator().manual_seed(0)
    for E, nk, nj in ((1, 1, 1), (7, 3, 4), (5000, 300, 12), (20000, 40, 5)):
        ii = torch.randint(0, nk, (E,), generator=g) * 3 + 100
        jj = torch.randint(0, nj, (E,), generator=g)
        rix, rjx = F.neighbors(ii, jj)
        ix, jx = fastba.neighbors(ii.to(DEV), jj.to(DEV))
        assert ix.dtype == torch.int64 and ix.is_cuda
        assert torch.equal(ix.cpu(), rix) and torch.equal(jx.cpu(), rjx)


def test_reproject_and_transform():
    from devo_amd.backends import cuda_ba
    poses, patches, intr, target, weight, ii, jj, kk = scene()
    d = lambda t: t.to(DEV)
    r = cuda_ba.reproject(d(poses), d(patches), d(intr), d(ii), d(jj), d(kk))
    assert_rel(r, F.reproject(poses, patches, intr, ii, jj, kk, dtype=torch.float64), 1e-5, "reproject")
    # push some patches behind / close to the camera to hit the Z clamps and validity gates
    patches2 = patches.clone()
    patches2[0, ::5, 2] = 40.0
    poses2 = poses.clone()
    poses2[0, 2, 2] -= 3.0
    O = OSE3(poses2.double())
    a64 = (patches2.double(), intr.double(), ii, jj, kk)
    c, v, (Ji, Jj, Jz) = pops.transform(O, *a64, jacobian=True)
    gc, gv, (gJi, gJj, gJz) = cuda_ba.transform(d(poses2), d(patches2), d(intr), d(ii), d(jj), d(kk), jacobian=True)
    assert_rel(gc, c, 1e-5, "coords")
    assert torch.equal(gv.cpu().double(), v)
    for got, ref, name in ((gJi, Ji, "Ji"), (gJj, Jj, "Jj"), (gJz, Jz, "Jz")):
        assert_rel(got, ref, 1e-4, name)
    assert_rel(cuda_ba.transform(d(poses2), d(patches2), d(intr), d(ii), d(jj), d(kk), depth=True), pops.transform(O, *a64, depth=True), 1e-5, "depth")
    assert_rel(cuda_ba.transform(d(poses2), d(patches2), d(intr), d(ii), d(jj), d(kk), tonly=True), pops.transform(O, *a64, tonly=True), 1e-5, "tonly")
    c2 = cuda_ba.transform(d(poses2), d(patches2), d(intr), d(ii), d(jj), d(kk), layout="2pp")
    assert torch.equal(c2, gc.permute(0, 1, 4, 2, 3).contiguous())       # devo.py:223


def test_transform_golden(golden_dir):
    """fused kernel vs outputs of the REAL reference projective_ops.transform (tests/golden)."""
    from devo_amd import projective_ops as P
    from devo_amd.lietorch import SE3
    z = np.load(os.path.join(golden_dir, "transform_f64.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim}
    d = lambda k: g[k].float().to(DEV) if g[k].is_floating_point() else g[k].to(DEV)
    args = (SE3(d("poses")), d("patches"), d("intrinsics"), d("ii"), d("jj"), d("kk"))
    with torch.no_grad():
        c, v, (Ji, Jj, Jz) = P.transform(*args, jacobian=True)
        fm = P.flow_mag(*args, beta=0.5)
    assert_rel(c, g["coords"], 1e-5, "coords")
    assert torch.equal(v.cpu(), g["valid"].float())
    assert_rel(Ji, g["Ji"], 1e-4, "Ji"); assert_rel(Jj, g["Jj"], 1e-4, "Jj"); assert_rel(Jz, g["Jz"], 1e-4, "Jz")
    assert_rel(fm, g["flow_mag"], 1e-4, "flow_mag")


def test_ba_empty_and_limits():
    from devo_amd.backends import cuda_ba
    s = scene()
    d = lambda t: t.to(DEV)
    P, Q = d(s[0]).clone(), d(s[1]).clone()
    e = torch.zeros(0, dtype=torch.long, device=DEV)
    assert cuda_ba.forward(P, Q, d(s[2]), d(s[3])[:, :0], d(s[4])[:, :0], torch.tensor([1e-4], device=DEV), e, e, e, 1, 8, 2) == []
    assert torch.equal(P.cpu(), s[0]) and torch.equal(Q.cpu(), s[1])          # no edges: nothing moves
    with pytest.raises(RuntimeError):                                           # more than 128 optimised poses
        big = torch.zeros(1, 140, 7, device=DEV); big[..., 6] = 1
        cuda_ba.forward(big, Q, d(s[2]), d(s[3]), d(s[4]), torch.tensor([1e-4], device=DEV), d(s[5]), d(s[6]), d(s[7]), 1, 140, 1)


def test_ba_failure_flag_on_breakdown():
    """NaN weights make the Cholesky pivot test fail: like the reference's exception (devo.py:336-340) the call
    leaves poses / patches untouched, and reports it through the device status flag."""
    from devo_amd.backends import cuda_ba
    s = scene()
    d = lambda t: t.to(DEV)
    P, Q = d(s[0]).clone(), d(s[1]).clone()
    w = d(s[4]).clone()
    w[0, 0, 0] = float("nan")
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    cuda_ba.forward(P, Q, d(s[2]), d(s[3]), w, torch.tensor([1e-4], device=DEV), d(s[5]), d(s[6]), d(s[7]), 1, 8, 2, status=status)
    assert int(status.item()) == 1
    assert torch.equal(P.cpu(), s[0]) and torch.equal(Q.cpu(), s[1])


def test_dropin_fastba_reports_a_failed_factorisation():
    """devo_amd.fastba.BA (the `devo.fastba.BA` drop-in): a Cholesky breakdown surfaces as an exception the caller's
    try / except sees (devo.py:336-340) — at the next call (lazy, no synchronisation in the failing call), immediately with
    check="now", or through last_status(); more than 128 optimised poses is a clear error."""
    from devo_amd import fastba
    s = scene()
    d = lambda t: t.to(DEV)
    args = lambda P, Q, w: (P, Q, d(s[2]), d(s[3]), w, torch.tensor([1e-4], device=DEV), d(s[5]), d(s[6]), d(s[7]), 1, 8, 2)
    bad = d(s[4]).clone(); bad[0, 0, 0] = float("nan")
    P, Q = d(s[0]).clone(), d(s[1]).clone()
    fastba.last_status(DEV)                                                     # (clear whatever earlier tests left)
    assert fastba.BA(*args(P, Q, d(s[4]))) == [] and fastba.last_status(DEV) == 0
    P, Q = d(s[0]).clone(), d(s[1]).clone()
    assert fastba.BA(*args(P, Q, bad)) == []                                    # fails silently for now ...
    torch.cuda.synchronize()
    P2, Q2 = d(s[0]).clone(), d(s[1]).clone()
    with pytest.raises(fastba.BAFailure):                                       # ... and is reported by the next call (the failed one has finished),
        fastba.BA(*args(P2, Q2, d(s[4])))
    Pr, Qr = d(s[0]).clone(), d(s[1]).clone()                                   # whose own (healthy) adjustment has still run
    assert fastba.BA(*args(Pr, Qr, d(s[4]))) == [] and fastba.last_status(DEV) == 0
    assert torch.equal(P2, Pr) and torch.equal(Q2, Qr) and not torch.equal(P2.cpu(), s[0])
    # without waiting for the GPU in between: by the next call or by the one after it (a loop never waits for the adjustment it just enqueued)
    assert fastba.BA(*args(d(s[0]).clone(), d(s[1]).clone(), bad)) == []
    raised = 0
    for _ in range(2):
        try:
            fastba.BA(*args(d(s[0]).clone(), d(s[1]).clone(), d(s[4])))
        except fastba.BAFailure:
            raised += 1
    assert raised == 1 and fastba.last_status(DEV) == 0
    with pytest.raises(fastba.BAFailure):
        fastba.BA(*args(d(s[0]).clone(), d(s[1]).clone(), bad), check="now")
    fastba.BA(*args(d(s[0]).clone(), d(s[1]).clone(), bad), check="never")
    assert fastba.last_status(DEV) == 1
    big = torch.zeros(1, 140, 7, device=DEV); big[..., 6] = 1
    with pytest.raises(RuntimeError, match="at most 128"):
        fastba.BA(big, Q, d(s[2]), d(s[3]), d(s[4]), torch.tensor([1e-4], device=DEV), d(s[5]), d(s[6]), d(s[7]), 1, 140, 1)


def test_prepare_then_forward_prepared_equals_forward():
    """cuda_ba.prepare + forward(prepared=True) is the same computation as forward(); a prepared workspace can be
    solved again (also after a run that broke down: the sticky failure flag is reset by the next call)."""
    from devo_amd.backends import cuda_ba
    poses, patches, intr, target, weight, ii, jj, kk = scene(seed=21)
    n = poses.shape[1]
    dev = lambda t: t.to(DEV)
    lm = torch.as_tensor([1e-4], device=DEV)
    args = (dev(intr), dev(target), dev(weight), lm, dev(ii), dev(jj), dev(kk), 1, n, 2)
    P0, Q0 = dev(poses.clone()), dev(patches.clone())
    cuda_ba.forward(P0, Q0, *args)
    E, Np = ii.numel(), patches.shape[1]
    ws = cuda_ba.workspace(E, Np, n - 1, torch.device(DEV))
    cuda_ba.prepare(dev(kk), Np, n - 1, ws)
    for _ in range(2):                                             # the same prepared workspace twice
        P1, Q1 = dev(poses.clone()), dev(patches.clone())
        cuda_ba.forward(P1, Q1, *args, ws=ws, prepared=True)
        assert torch.equal(P0, P1) and torch.equal(Q0, Q1)
    # break it (NaN target -> Cholesky breakdown), then solve again on the same workspace
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    bad = list(args); bad[1] = torch.full_like(args[1], float("nan"))
    P2, Q2 = dev(poses.clone()), dev(patches.clone())
    cuda_ba.forward(P2, Q2, *bad, ws=ws, prepared=True, status=status)
    assert int(status) > 0
    P3, Q3 = dev(poses.clone()), dev(patches.clone())
    cuda_ba.forward(P3, Q3, *args, ws=ws, prepared=True, status=status)
    assert int(status) == 0 and torch.equal(P0, P3) and torch.equal(Q0, Q3)
    with pytest.raises(RuntimeError):
        cuda_ba.forward(P3, Q3, *args, prepared=True)              # needs the prepared workspace
    # a workspace that was never prepared (or prepared for another graph size) is refused on the device: nothing is
    # touched and the status flag says -1 — no walk through garbage tables
    for junk in (torch.zeros_like(ws), torch.randint(0, 255, ws.shape, dtype=torch.uint8, device=DEV)):
        P4, Q4 = dev(poses.clone()), dev(patches.clone())
        status.zero_()
        cuda_ba.forward(P4, Q4, *args, ws=junk, prepared=True, status=status)
        torch.cuda.synchronize()
        assert int(status) == -1 and torch.equal(P4, dev(poses)) and torch.equal(Q4, dev(patches))


def test_update_schedule_18_iterations():
    """the whole update schedule of enet.py:300-339 — 18 update iterations on a growing graph (8 frames
    for the first 8 iterations, then one more frame per iteration up to 15), each = reproject -> target = centre +
    replayed delta -> 2 Gauss-Newton iterations — must stay within 1e-4 of the fp64 oracle run on the same replay."""
    from devo_amd.backends import cuda_ba
    n, M, H, W = 15, 16, 120, 160
    poses = synth.make_poses(n, 31)
    patches, _ = synth.make_patches(n, M, H, W, seed=31)
    intr = synth.make_intrinsics(n, H, W)
    dev = lambda t: t.to(DEV)
    P, Q = dev(poses.clone()), dev(patches.clone())
    P64, Q64 = poses.double(), patches.double()
    lm = torch.tensor([1e-4])
    ws = None
    for it in range(18):
        nk = 8 if it < 8 else min(15, 8 + it - 7)
        ii, jj, kk = synth.full_graph(nk, M)
        delta, weight = synth.make_update_outputs(len(ii), 100 + it, sigma=0.3)
        # GPU path
        c = cuda_ba.transform(P, Q, dev(intr), dev(ii), dev(jj), dev(kk), layout="2pp")
        target = c[:, :, :, 1, 1] + dev(delta)
        cuda_ba.forward(P, Q, dev(intr), target, dev(weight), dev(lm), dev(ii), dev(jj), dev(kk), 1, nk, 2)
        # fp64 oracle on the same replay
        c64 = pops.transform(OSE3(P64), Q64, intr.double(), ii, jj, kk)
        t64 = c64[..., 1, 1, :] + delta.double()
        P64, Q64 = F.ba(P64, Q64, intr.double(), t64, weight.double(), lm, ii, jj, kk, 1, nk, 2, dtype=torch.float64)
    check((P.cpu(), Q.cpu()), (P64, Q64), tol=1e-4)


@pytest.mark.parametrize("E,Np,ordered", [(5000, 300, False), (21600, 1440, True), (7013, 900, "ragged"), (30001, 2000, "ragged"),
                                              (40000, 700, False), (200000, 5000, False)])
def test_prepare_tables_bit_exact(E, Np, ordered):
    """The index half of the BA against torch.unique(kk, sorted, return_inverse) (ba_cuda.cu:435-437), bit-exact:
    sorted unique patch ids and, per patch, exactly its edges in ascending order — for the register-cached
    single-workgroup kernel (E <= 32768), the re-reading one and the multi-kernel path (E > 2^17)."""
    from devo_amd.backends import cuda_ba
    g = torch.Generator().manual_seed(E)
    if ordered == "ragged":                                     # ascending ids, runs of any length, ids missing: the run-head path
        kk = torch.sort(torch.randint(0, Np, (E,), generator=g) // 3 * 3).values
    elif ordered:
        kk = torch.arange(Np).repeat_interleave(E // Np)
    else:
        kk = torch.randint(0, Np, (E,), generator=g)
        kk[kk % 7 == 3] = 5                                     # gaps in the id range and one very long segment
    ws = cuda_ba.workspace(E, Np, 7, torch.device(DEV))
    cuda_ba.prepare(kk.to(DEV), Np, 7, ws)
    n_seg, kx, seg, perm = cuda_ba.prepared_tables(ws, E, Np, 7)
    kx_ref, inv = torch.unique(kk, sorted=True, return_inverse=True)
    assert n_seg == len(kx_ref) and torch.equal(kx.cpu().long(), kx_ref)
    seg, perm = seg.cpu().long(), perm.cpu().long()
    assert seg[0] == 0 and seg[-1] == E and torch.equal(seg[1:] - seg[:-1], torch.bincount(inv, minlength=n_seg))
    assert torch.equal(inv[perm], torch.repeat_interleave(torch.arange(n_seg), seg[1:] - seg[:-1]))   # grouped by patch
    same = inv[perm][1:] == inv[perm][:-1]
    assert bool((perm[1:][same] > perm[:-1][same]).all())       # ascending edge ids inside every patch
    assert torch.equal(torch.sort(perm).values, torch.arange(E))


@pytest.mark.parametrize("seed", range(6))
def test_ba_random_graphs(seed):
    """random graphs: frame counts 2..16, few patches, dropped / duplicated / shuffled edges, different fixed-pose
    windows — the register kernels of every size class (<= 8, 11, 14, 16 poses) and the irregular-patch path"""
    g = torch.Generator().manual_seed(500 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    n = [3, 6, 9, 12, 15, 17][seed]
    s = list(scene(n=n, M=ri(2, 7), H=96, W=128, seed=40 + seed, keep=[1.0, 0.8, 0.6][ri(0, 2)], sigma=0.5))
    if seed % 2:                                               # duplicate a few edges (irregular patches)
        E = len(s[5])
        dup = torch.randint(0, E, (max(2, E // 20),), generator=g)
        s[3] = torch.cat([s[3], s[3][:, dup]], 1); s[4] = torch.cat([s[4], s[4][:, dup]], 1)
        for k in (5, 6, 7):
            s[k] = torch.cat([s[k], s[k][dup]])
    t0 = ri(1, max(1, n - 2))
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], t0, n, 2, dtype=torch.float64)
    check(run_ba(*s, t0, n, 2), ref)


@pytest.mark.parametrize("n,M,t0", [(8, 300, 1), (11, 210, 2), (14, 150, 1), (16, 80, 1), (16, 80, 0)])
def test_ba_with_more_patches_than_accumulate_waves(n, M, t0):
    """the accumulate kernel keeps one copy of the block triangle per wave (LDS slab); with more patches than the launch has waves
    (256 workgroups x 8 / 6 / 4 waves for N <= 11 / 14 / 16) a wave folds a second patch into a slab that is no longer fresh —
    every size class, against the fp64 oracle; sigma 0.3 px keeps fp32 rounding inside 1e-4 at these sizes"""
    s = scene(n=n, M=M, H=96, W=128, seed=70 + n, sigma=0.3, keep=0.9)
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], t0, n, 2, dtype=torch.float64)
    check(run_ba(*s, t0, n, 2), ref)


@pytest.mark.parametrize("n", [22, 28])
def test_ba_more_than_16_optimised_poses_uses_the_general_kernel(n):
    """N = t1 - t0 in 17..32: the LDS-atomic accumulate kernel (no register-resident S); the solve's back substitution keeps
    z in registers up to 6 N = 128 (n = 22: 126 rows, both register rows in use) and in LDS beyond (n = 28)."""
    s = scene(n=n, M=4, H=96, W=128, seed=77, keep=0.9, sigma=0.5)
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], 1, n, 2, dtype=torch.float64)
    check(run_ba(*s, 1, n, 2), ref)


@pytest.mark.parametrize("n,t0", [(34, 1), (40, 0), (57, 2)])
def test_ba_more_than_32_optimised_poses_keeps_the_system_in_global_memory(n, t0):
    """The reference has no limit on t1 - t0 (ba_cuda.cu:516-522: dense S, torch::linalg::cholesky).  Beyond 32 optimised poses the
    reduced system does not fit one workgroup's LDS: accumulate with device-scope atomics into the global image (like the reference's
    atomicAdds), Schur term as one product, the blocked Cholesky in place on the global image.  Same tolerance as every other BA."""
    s = scene(n=n, M=4, H=96, W=128, seed=31 + n, keep=0.9, sigma=0.5)
    ref = F.ba(*[x.double() if x.is_floating_point() else x for x in s[:5]], torch.tensor([1e-4]), *s[5:], t0, n, 2, dtype=torch.float64)
    ref32 = F.ba(*s[:5], torch.tensor([1e-4]), *s[5:], t0, n, 2, dtype=torch.float32)
    check(run_ba(*s, t0, n, 2), ref, ref32=ref32)


def test_prepare_with_plan_equals_the_two_separate_calls():
    """cuda_ba.prepare(..., plan=...) (BA index preparation + the lookup plan's ordering step as two workgroups of one launch)
    leaves the same BA tables and a plan with the same heavy set and the same (frame, band) sequence as the separate calls"""
    from devo_amd.backends import cuda_ba, cuda_corr
    n, M, H, W = 6, 40, 60, 80
    poses, (patches, _), intr = synth.make_poses(n, 9), synth.make_patches(n, M, H, W, seed=9), synth.make_intrinsics(n, H, W)
    ii, jj, kk = (t.to(DEV) for t in synth.full_graph(n, M))
    E, Np = ii.numel(), patches.shape[1]
    args = (poses.to(DEV), patches.to(DEV), intr.to(DEV), ii, jj, kk)
    coords, buf_a = cuda_ba.transform(*args, layout="2pp", plan_for=(n, H, 3))
    _, buf_b = cuda_ba.transform(*args, layout="2pp", plan_for=(n, H, 3))
    ws_a, ws_b = cuda_ba.workspace(E, Np, n - 1, DEV), cuda_ba.workspace(E, Np, n - 1, DEV)
    cuda_ba.prepare(kk, Np, n - 1, ws_a, plan=(buf_a, n, H))
    cuda_ba.prepare(kk, Np, n - 1, ws_b)
    plan_b = cuda_corr.plan_finish(buf_b, jj, n, H, 3)
    ta, tb = cuda_ba.prepared_tables(ws_a, E, Np, n - 1), cuda_ba.prepared_tables(ws_b, E, Np, n - 1)
    assert ta[0] == tb[0] and all(torch.equal(x, y) for x, y in zip(ta[1:], tb[1:]))
    a, b = buf_a.cpu(), plan_b.cpu()
    nh = int(a[E])
    assert nh == int(b[E]) and sorted(a[:E].tolist()) == list(range(E)) and sorted(a[:nh].tolist()) == sorted(b[:nh].tolist())
    key = lambda o: torch.stack([jj.cpu()[o[nh:E].long()], (coords.cpu()[0, o[nh:E].long(), 1, 1, 1].clamp(0, H - 1) / 16).floor().long()], 1)
    assert torch.equal(key(a), key(b))
    # and the BA runs on the tables of the combined launch
    tgt = coords[:, :, :, 1, 1] + 0.1
    w = torch.ones(1, E, 2, device=DEV)
    pa, qa, pb, qb = args[0].clone(), args[1].clone(), args[0].clone(), args[1].clone()
    lm = torch.tensor([1e-4], device=DEV)
    cuda_ba.forward(pa, qa, args[2], tgt, w, lm, ii, jj, kk, 1, n, 2, ws=ws_a, prepared=True)
    cuda_ba.forward(pb, qb, args[2], tgt, w, lm, ii, jj, kk, 1, n, 2)
    assert torch.equal(pa, pb) and torch.equal(qa, qb)


@pytest.mark.parametrize("layout", ["2pp", "pp2"])
def test_forward_delta_equals_forward_on_the_formed_target(layout):
    """devo.py:330 folded into the BA (devo_ba_forward_prepared_delta): bit-identical poses and patches to forming
    target = coords[..., 1, 1] + delta with torch first."""
    from devo_amd.backends import cuda_ba
    nk, M, H, W = 9, 40, 120, 160
    poses = synth.make_poses(nk, 5)
    patches, _ = synth.make_patches(nk, M, H, W, seed=5)
    intr = synth.make_intrinsics(nk, H, W)
    dev = lambda t: t.to(DEV)
    ii, jj, kk = synth.full_graph(nk, M)
    E = len(ii)
    delta, weight = synth.make_update_outputs(E, 7, sigma=0.4)
    lm = torch.tensor([1e-4])
    Np = patches.shape[1] if patches.dim() == 5 else patches.shape[0]
    outs = []
    for fused in (False, True):
        P_, Q_ = dev(poses.clone()), dev(patches.clone())
        ws = cuda_ba.workspace(E, Np, nk - 1, torch.device(DEV))
        cuda_ba.prepare(dev(kk), Np, nk - 1, ws)
        c = cuda_ba.transform(P_, Q_, dev(intr), dev(ii), dev(jj), dev(kk), layout=layout)
        if fused:
            cuda_ba.forward_delta(P_, Q_, dev(intr), c, dev(delta), dev(weight), dev(lm), dev(ii), dev(jj), dev(kk), 1, nk, 2, ws,
                                  layout=layout)
        else:
            centre = c[:, :, :, 1, 1] if layout == "2pp" else c[:, :, 1, 1, :2]
            cuda_ba.forward(P_, Q_, dev(intr), centre + dev(delta), dev(weight), dev(lm), dev(ii), dev(jj), dev(kk), 1, nk, 2,
                            ws=ws, prepared=True)
        outs.append((P_.cpu(), Q_.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], poses)


def test_ba_is_bit_reproducible_on_regular_graphs():
    """Up to 16 optimised poses and regular patches: fixed summation orders, no floating-point atomics — repeated calls on the
    same inputs return the same bits (the reference's global atomicAdds, ba_cuda.cu:297-322, do not)."""
    from devo_amd.backends import cuda_ba
    nk, M, H, W = 12, 48, 120, 160
    poses = synth.make_poses(nk, 9).to(DEV)
    patches = synth.make_patches(nk, M, H, W, seed=9)[0].to(DEV)
    intr = synth.make_intrinsics(nk, H, W).to(DEV)
    ii, jj, kk = [t.to(DEV) for t in synth.full_graph(nk, M)]
    delta, weight = [t.to(DEV) for t in synth.make_update_outputs(len(ii), 9, sigma=0.3)]
    lm = torch.tensor([1e-4], device=DEV)
    c = cuda_ba.transform(poses, patches, intr, ii, jj, kk, layout="2pp")
    target = c[:, :, :, 1, 1] + delta
    first = None
    for _ in range(25):
        P_, Q_ = poses.clone(), patches.clone()
        cuda_ba.forward(P_, Q_, intr, target, weight, lm, ii, jj, kk, 1, nk, 2)
        if first is None:
            first = (P_, Q_)
            assert not torch.equal(P_, poses)
        else:
            assert torch.equal(P_, first[0]) and torch.equal(Q_, first[1])


_SOLVER_AB = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from devo_amd import synth
from devo_amd.backends import cuda_ba
out = {}
for nk, M, t0 in ((2, 40, 1), (3, 40, 1), (6, 64, 1), (11, 48, 1), (12, 48, 1), (14, 80, 1), (15, 96, 1), (17, 40, 1), (20, 30, 1), (22, 30, 1)):
    poses = synth.make_poses(nk, nk).cuda(); patches = synth.make_patches(nk, M, 120, 160, seed=nk)[0].cuda()
    intr = synth.make_intrinsics(nk, 120, 160).cuda()
    ii, jj, kk = [t.cuda() for t in synth.full_graph(nk, M)]
    delta, weight = [t.cuda() for t in synth.make_update_outputs(len(ii), nk, sigma=0.3)]
    c = cuda_ba.transform(poses, patches, intr, ii, jj, kk, layout="2pp")
    cuda_ba.forward(poses, patches, intr, c[:, :, :, 1, 1] + delta, weight, torch.tensor([1e-4]).cuda(), ii, jj, kk, t0, nk, 2)
    out[f"{nk}_{t0}"] = (poses.cpu(), patches.cpu())
torch.save(out, sys.argv[2])
"""


def test_the_one_barrier_solver_returns_the_bits_of_the_two_barrier_form(tmp_path):
    """k_ba_solve_chain (one barrier per block step, panel solved inside the tile waves, 6 N <= 128) performs the operations of k_ba_solve
    in the same order: the same bits for 1 .. 16 optimised poses (more tiles than tile waves included; the back-substitution's unrolled chain
    entered at every depth), agreement for 19 and 21.  DEVO_BA_SOLVE_V1 is read
    once per process: both forms run in sub-processes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for v1 in (False, True):
        env = dict(os.environ)
        env.pop("DEVO_BA_SOLVE_V1", None)
        if v1:
            env["DEVO_BA_SOLVE_V1"] = "1"
        path = str(tmp_path / f"solver_{int(v1)}.pt")
        r = subprocess.run([sys.executable, "-c", _SOLVER_AB, root, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        res.append(torch.load(path))
    assert res[0].keys() == res[1].keys()
    for k in res[0]:
        assert torch.isfinite(res[0][k][0]).all()
        if int(k.split("_")[0]) - int(k.split("_")[1]) <= 16:
            assert torch.equal(res[0][k][0], res[1][k][0]) and torch.equal(res[0][k][1], res[1][k][1]), k
        else:         # more than 16 optimised poses: the general accumulate kernel adds with atomics, the system itself differs in the last bits
            assert torch.allclose(res[0][k][0], res[1][k][0], atol=1e-4) and torch.allclose(res[0][k][1], res[1][k][1], atol=1e-4), k
