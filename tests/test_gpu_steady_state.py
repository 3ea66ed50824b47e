"""The shape DEVO's INFERENCE runs (config/default.yaml:4-7, devo/devo.py:69,210-217,320-337,366-380): the sliding-window patch graph after
40 keyframes — 96 patches per frame, PATCH_LIFETIME 13, REMOVAL_WINDOW 22: 45 312 edges, kk unsorted — the ring buffers of mem = 32 frames
addressed modulo the ring (`kk % (M * mem)`, `jj % mem`: frames 6, 7 of the graph alias slots of frames 38, 39, as in the reference), the
optimisation window t0 = n - 10, and the ring written ONE slot per frame.  Lookup against the oracle on a stratified sample (wrapped /
unwrapped indices), the BA against the fp64 oracle in full, `neighbors` bit-exact, and the per-slot maintenance of the converted ring
(devo_amd/backends/ring.py) against a conversion of the whole ring: the same bits, one frame converted per written slot."""
import pytest
import torch
from oracle import altcorr as A
from oracle import fastba as F
from util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_KF, M, MEM, H, W, C, R = 40, 96, 32, 120, 160, 128, 3


def _scene(seed=11):
    from devo_amd import synth
    nbuf = 48                                                          # (the reference's pose / patch buffers hold 2048 frames: only their size differs)
    # a camera that moves like a camera: ~1 cm and 0.1 degree per keyframe, so that the 13 frames a patch is tracked through overlap (86 % of
    # the reprojections land inside the frame, every patch keeps >= 13 unmasked edges).  synth.make_poses' defaults at n = 40 scatter the frames
    # (patches whose only unmasked edge is the one into their own frame: depth unobservable, dz = fp32 noise / lambda in any implementation)
    poses = synth.make_poses(nbuf, seed, trans_step=0.01, rot_step=0.002)
    patches, centres = synth.make_patches(nbuf, M, H, W, seed=seed)
    intr = synth.make_intrinsics(nbuf, H, W)
    ii, jj, kk = synth.sliding_window_graph(N_KF, M)
    assert len(ii) == 45312 and int(jj.max()) >= MEM and int(kk.max()) >= M * MEM and not bool((kk[1:] >= kk[:-1]).all())
    return poses, patches, centres, intr, ii, jj, kk


def _ring(centres, dtype, seed=11):
    """The ring as devo.py:523-527 leaves it after N_KF frames: slot f % MEM holds the LAST frame written there."""
    from devo_amd import synth
    fmap, gmap = synth.make_features(N_KF, M, C, H, W, centres[:N_KF], seed=seed)
    f1 = synth.pyramid_l1(fmap)
    fmap1_ = torch.zeros(1, MEM, C, H, W, dtype=dtype)
    fmap2_ = torch.zeros(1, MEM, C, H // 4, W // 4, dtype=dtype)
    gmap_ = torch.zeros(MEM, M, C, 3, 3, dtype=dtype)
    g5 = gmap.view(N_KF, M, C, 3, 3)
    for f in range(N_KF):
        fmap1_[:, f % MEM] = fmap[:, f].to(dtype)
        fmap2_[:, f % MEM] = f1[:, f].to(dtype)
        gmap_[f % MEM] = g5[f].to(dtype)
    return fmap1_, fmap2_, gmap_, (fmap, f1, g5)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float16, 2e-3)])
def test_lookup_into_the_wrapped_ring(dtype, tol):
    """DEVO.corr (devo.py:210-217) at the steady-state shape through the reference's own call: two altcorr.corr calls with indices modulo
    the ring + torch.stack; the fused pyramid call writes the same bits; oracle on 256 edges, half of them with a wrapped index."""
    from devo_amd import altcorr
    from devo_amd.backends import cuda_ba
    poses, patches, centres, intr, ii, jj, kk = _scene()
    fmap1_, fmap2_, gmap_, _ = _ring(centres, dtype)
    d = lambda t: t.to(DEV)
    coords = cuda_ba.transform(d(poses), d(patches), d(intr), d(ii), d(jj), d(kk), layout="2pp")
    pyr = (d(fmap1_), d(fmap2_))
    gm = d(gmap_).view(1, MEM * M, C, 3, 3)
    ii1, jj1 = d(kk) % (M * MEM), d(jj) % MEM                        # devo.py:213-214
    E = len(ii)
    corr1 = altcorr.corr(gm, pyr[0], coords / 1, ii1, jj1, R)
    corr2 = altcorr.corr(gm, pyr[1], coords / 4, ii1, jj1, R)
    corr = torch.stack([corr1, corr2], -1).view(1, E, -1)
    assert bool(torch.isfinite(corr).all())
    fused = altcorr.corr_pyramid(gm, list(pyr), coords, ii1, jj1, radius=R, scales=(1, 4))
    assert torch.equal(fused.view(1, E, -1), corr)
    wrapped = ((kk >= M * MEM) | (jj >= MEM)).nonzero().squeeze(1)
    plain = ((kk < M * MEM) & (jj < MEM)).nonzero().squeeze(1)
    g = torch.Generator().manual_seed(5)
    sel = torch.cat([wrapped[torch.randperm(len(wrapped), generator=g)[:128]], plain[torch.randperm(len(plain), generator=g)[:128]]])
    assert len(wrapped) > 1000 and len(sel) == 256
    q = (lambda t: t.float()) if dtype == torch.float16 else (lambda t: t)
    c_cpu = coords.cpu()[:, sel]
    k1, j1 = (kk % (M * MEM))[sel], (jj % MEM)[sel]
    gm_cpu = q(gmap_).view(1, MEM * M, C, 3, 3)
    ref = torch.stack([A.corr_forward(gm_cpu, q(fmap1_), c_cpu, k1, j1, R), A.corr_forward(gm_cpu, q(fmap2_), c_cpu / 4, k1, j1, R)], -1).reshape(1, 256, -1)
    assert rel_err(corr.cpu()[:, sel].float(), ref) <= tol


def test_bundle_adjustment_over_the_optimisation_window():
    """fastba.BA as devo.py:320-337 calls it in steady state: 45 312 edges in DEVO's order, t0 = n - OPTIMIZATION_WINDOW (10 optimised poses
    out of 48 in the buffer, patches of 22 frames), 2 iterations — fp64 oracle in full, 1e-4 per tensor; kk is not sorted: the general index
    preparation; a second call on the same kk takes the remembered tables and returns the same bits."""
    from devo_amd import synth
    from devo_amd.backends import cuda_ba
    poses, patches, centres, intr, ii, jj, kk = _scene()
    d = lambda t: t.to(DEV)
    P_, Q_ = d(poses).clone(), d(patches).clone()
    coords = cuda_ba.transform(P_, Q_, d(intr), d(ii), d(jj), d(kk), layout="2pp")
    delta, weight = synth.make_update_outputs(len(ii), 11, sigma=0.5)
    target = coords[:, :, :, 1, 1] + d(delta)
    lm = torch.tensor([1e-4], device=DEV)
    t0, t1 = N_KF - 10, N_KF
    gi, gj, gk = d(ii), d(jj), d(kk)
    cuda_ba.forward(P_, Q_, d(intr), target, d(weight), lm, gi, gj, gk, t0, t1, 2)
    assert cuda_ba.last_path() == "accumulate:register solve:chain"
    pr, qr = F.ba(poses.double(), patches.double(), intr.double(), target.cpu().double(), weight.double(), torch.tensor([1e-4]), ii, jj, kk, t0, t1, 2,
                  dtype=torch.float64)
    assert not torch.equal(P_.cpu(), poses)
    assert torch.equal(P_.cpu()[:, :t0], poses[:, :t0]) and torch.equal(P_.cpu()[:, t1:], poses[:, t1:])       # fixed poses untouched
    assert rel_err(P_.cpu()[..., :3], pr[..., :3]) <= 1e-4 and rel_err(P_.cpu()[..., 3:], pr[..., 3:]) <= 1e-4
    assert rel_err(Q_.cpu()[:, :, 2], qr[:, :, 2]) <= 1e-4
    assert torch.equal(Q_.cpu()[:, :, :2], patches[:, :, :2])
    h, m = cuda_ba.prep_stats()
    P2, Q2 = d(poses).clone(), d(patches).clone()
    cuda_ba.forward(P2, Q2, d(intr), target, d(weight), lm, gi, gj, gk, t0, t1, 2)
    assert cuda_ba.prep_stats() == (h + 1, m)
    assert torch.allclose(P2, P_, atol=1e-6) and torch.allclose(Q2, Q_, atol=1e-6)


def test_neighbors_of_the_sliding_window_graph_bit_exact():
    """fastba.neighbors(kk, jj) as the update operator asks for it (enet.py:80-81) on DEVO's unsorted lists."""
    from devo_amd.backends import cuda_ba
    _, _, _, _, ii, jj, kk = _scene()
    ix, jx = cuda_ba.neighbors(kk.to(DEV), jj.to(DEV))
    rx, ry = F.neighbors(kk, jj)
    assert torch.equal(ix.cpu(), rx) and torch.equal(jx.cpu(), ry)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_one_written_slot_converts_one_frame(dtype):
    """devo.py:523-527 and :288-291: frames arrive one at a time, each written into ONE slot of the rings (`fmap1_[:, k] = ...`,
    `gmap_[k] = ...`); keyframe removal copies a few slots.  With devo_amd.backends.install() every such write is on record and the next
    lookup converts just the written frames / patches — the result is bit for bit the lookup into a ring converted as a whole, and the
    conversion counters say what was converted.  An in-place operation the wrapper does not see (`mul_`) converts everything again."""
    import devo_amd.backends as B
    from devo_amd.backends import cuda_ba, ring
    from devo_amd import altcorr
    from devo_amd.backends import cuda_corr as CC
    mods = B.install()
    assert ring.tracking()                                             # (both bindings keep write records since round 6: DEVO_BINDING=ctypes runs this test too)
    poses, patches, centres, intr, ii, jj, kk = _scene()
    _, _, _, (fmap, f1, g5) = _ring(centres, dtype)
    d = lambda t: t.to(DEV)
    coords = cuda_ba.transform(d(poses), d(patches), d(intr), d(ii), d(jj), d(kk), layout="2pp")
    ii1, jj1 = d(kk) % (M * MEM), d(jj) % MEM
    fmap1_ = torch.zeros(1, MEM, C, H, W, dtype=dtype, device=DEV)
    fmap2_ = torch.zeros(1, MEM, C, H // 4, W // 4, dtype=dtype, device=DEV)
    gmap_ = torch.zeros(MEM, M, C, 3, 3, dtype=dtype, device=DEV)
    fmap, f1, g5 = d(fmap).to(dtype), d(f1).to(dtype), d(g5).to(dtype)

    def lookup():
        gm = gmap_.view(1, MEM * M, C, 3, 3)
        return torch.stack([altcorr.corr(gm, fmap1_, coords / 1, ii1, jj1, R), altcorr.corr(gm, fmap2_, coords / 4, ii1, jj1, R)], -1)

    def whole_ring_reference():
        CC.clear_caches()
        ref = lookup()
        return ref

    for f in range(MEM):                                               # fill the ring, then look it up once: everything is converted
        fmap1_[:, f % MEM] = fmap[:, f]; fmap2_[:, f % MEM] = f1[:, f]; gmap_[f % MEM] = g5[f]
    CC.clear_caches()
    lookup()
    s0 = CC.convert_stats()
    for f in range(MEM, N_KF):                                         # steady state: one new frame, one lookup
        k = f % MEM
        gmap_[k] = g5[f]                                               # devo.py:524
        fmap1_[:, k] = fmap[:, f]                                      # devo.py:526
        fmap2_[:, k] = f1[:, f]                                        # devo.py:527
        got = lookup()
        s1 = CC.convert_stats()
        assert s1[0] == s0[0] and s1[2] == s0[2], f"frame {f}: a whole tensor was converted again {s0} -> {s1}"
        assert s1[1] - s0[1] == 2 and s1[3] - s0[3] == 1 and s1[4] - s0[4] == M, (s0, s1)      # one frame per level, one patch range of M patches
        s0 = s1
        if f in (MEM, N_KF - 1):
            assert torch.equal(got, whole_ring_reference())
            lookup()
            s0 = CC.convert_stats()
    # keyframe removal (devo.py:288-291): slots i <- i + 1 for a few i, through integer indices of batch entry 0
    for i in range(35, 39):
        gmap_[i % MEM] = gmap_[(i + 1) % MEM]
        fmap1_[0, i % MEM] = fmap1_[0, (i + 1) % MEM]
        fmap2_[0, i % MEM] = fmap2_[0, (i + 1) % MEM]
    got = lookup()
    s1 = CC.convert_stats()
    assert s1[0] == s0[0] and s1[2] == s0[2] and s1[1] - s0[1] == 8 and s1[4] - s0[4] == 4 * M, (s0, s1)
    assert torch.equal(got, whole_ring_reference())
    lookup()
    s0 = CC.convert_stats()
    # a write the wrapper cannot place, and an in-place operation it never sees: the whole tensor again (never a stale slot)
    fmap1_[:, torch.tensor([3, 7], device=DEV)] = fmap[:, :2]          # advanced indexing: no single contiguous range
    fmap2_.mul_(0.5)
    got = lookup()
    s1 = CC.convert_stats()
    assert s1[0] - s0[0] == 2 and s1[1] == s0[1], (s0, s1)
    assert torch.equal(got, whole_ring_reference())


@pytest.mark.parametrize("graph", ["sliding", "full", "small", "big_random"])
def test_graph_tables_in_one_call_equal_the_separate_calls(graph):
    """devo_upd_graph_tables (round 6: the Update operator's neighbours + patch groups + frame-pair groups of a NEW edge list from one library call —
    per-frame work in DEVO's steady state) against the separate entry points it replaces: cuda_ba.neighbors (ba.cpp:104-149), cuda_ba.prepare on kk,
    and the groups of ii * 12345 + jj (enet.py:94) from torch.unique.  Bit for bit; the frame-pair groups hold >= 64 edges each in the sliding-window
    graph, which takes the LDS-counted rank / scatter passes of the preparation."""
    from devo_amd import synth
    from devo_amd.backends import cuda_ba
    dev = "cuda"
    if graph == "sliding":
        ii, jj, kk = [t.to(dev) for t in synth.sliding_window_graph(40, 96)]
    elif graph == "full":
        ii, jj, kk = [t.to(dev) for t in synth.full_graph(15, 96)]
    elif graph == "big_random":                                        # beyond the single-workgroup preparation: both lists through Prep2, ragged groups
        g = torch.Generator().manual_seed(5)
        kk = torch.randint(0, 3000, (50001,), generator=g).to(dev)
        ii = kk // 100
        jj = torch.randint(0, 40, (50001,), generator=g).to(dev)
    else:
        g = torch.Generator().manual_seed(3)
        kk = torch.randint(0, 300, (2500,), generator=g).to(dev)
        ii = kk // 20
        jj = torch.randint(0, 15, (2500,), generator=g).to(dev)
    E = kk.numel()
    ix, jx, tk, tp = cuda_ba.graph_tables(ii, jj, kk)
    ix0, jx0 = cuda_ba.neighbors(kk, jj)
    assert torch.equal(ix, ix0) and torch.equal(jx, jx0)
    ws = cuda_ba.workspace(E, 1 << 20, 0, dev)
    cuda_ba.prepare(kk, 1 << 20, 0, ws)
    n, kx, seg, perm = cuda_ba.prepared_tables(ws, E, 1 << 20, 0)
    assert int(tk[0]) == n and torch.equal(tk[1][:n + 1], seg) and torch.equal(tk[2], perm)
    # frame-pair groups: the same partition of the edges as torch.unique's, groups in ascending (ii, jj) order, edges ascending inside a group
    key = ii * 12345 + jj
    uq, inv = torch.unique(key, return_inverse=True)
    m = int(tp[0])
    assert m == uq.numel()
    segp, permp = tp[1][:m + 1].long(), tp[2].long()
    assert int(segp[0]) == 0 and int(segp[-1]) == E
    grp = torch.repeat_interleave(torch.arange(m, device=dev), segp[1:] - segp[:-1])
    assert torch.equal(inv[permp], grp)
    inside = (permp[1:] > permp[:-1]) | (grp[1:] != grp[:-1])
    assert bool(inside.all())
    assert torch.equal(torch.sort(permp).values, torch.arange(E, device=dev))
