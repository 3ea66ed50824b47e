"""devo.py:311,337 hand the same kk to the Update operator and, right behind it, to fastba.BA: the operator's patch-group tables
(cuda_ba.graph_tables) serve the BA through ONE launch (devo_ba_import_tables) instead of a second preparation (nine launches at the
steady-state graph's 45 312 edges).  Everything here is an equality of bits with the BA that prepares its own tables, and the ways the
shortcut must NOT be taken."""
import pytest
import torch
from devo_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
N_KF, M = 40, 96


def _scene(seed=11, nkf=N_KF):
    nbuf = 48
    poses = synth.make_poses(nbuf, seed, trans_step=0.01, rot_step=0.002).to(DEV)
    patches = synth.make_patches(nbuf, M, 120, 160, seed=seed)[0].to(DEV)
    intr = synth.make_intrinsics(nbuf, 120, 160).to(DEV)
    ii, jj, kk = [t.to(DEV) for t in synth.sliding_window_graph(nkf, M)]
    delta, weight = [t.to(DEV) for t in synth.make_update_outputs(len(ii), seed, sigma=0.5)]
    from devo_amd.backends import cuda_ba
    c = cuda_ba.transform(poses, patches, intr, ii, jj, kk, layout="2pp")
    return poses, patches, intr, c[:, :, :, 1, 1] + delta, weight, torch.tensor([1e-4], device=DEV), ii, jj, kk


def _ba(scene, kk=None, t0=N_KF - 10, t1=N_KF):
    from devo_amd import fastba
    poses, patches, intr, tgt, w, lm, ii, jj, kk0 = scene
    P, Q = poses.clone(), patches.clone()
    fastba.BA(P, Q, intr, tgt, w, lm, ii, jj, kk0 if kk is None else kk, t0, t1, 2, check="now")
    return P, Q


def test_the_ba_takes_the_update_operators_tables_and_changes_no_bit():
    from devo_amd.backends import cuda_ba
    from devo_amd import fastba
    sc = _scene()
    ii, jj, kk = sc[6:]
    E, Np = kk.numel(), sc[1].shape[1]
    ref = _ba(sc)                                                       # (no donor yet: the BA's own preparation)
    n0 = cuda_ba.import_stats()
    cuda_ba.graph_tables(ii, jj, kk)                                    # what Update._tables does for a new graph
    got = _ba(sc)
    assert cuda_ba.import_stats() == n0 + 1
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]) and not torch.equal(got[0], sc[0])
    # the imported tables are the ones devo_ba_prepare builds
    ws = fastba._workspace(E, Np, 10, kk.device)
    imp = cuda_ba.prepared_tables(ws, E, Np, 10)
    ws2 = cuda_ba.workspace(E, Np, 10, kk.device)
    cuda_ba.prepare(kk, Np, 10, ws2)
    own = cuda_ba.prepared_tables(ws2, E, Np, 10)
    assert imp[0] == own[0] and all(torch.equal(a, b) for a, b in zip(imp[1:], own[1:]))
    # a second adjustment of the same graph (the next update iteration) imports nothing: the workspace holds the tables
    # (prepare() above was an explicit preparation of ANOTHER workspace: it forgets what was imported — once more, then)
    _ba(sc)
    n1 = cuda_ba.import_stats()
    again = _ba(sc)
    assert cuda_ba.import_stats() == n1 and torch.equal(ref[0], again[0]) and torch.equal(ref[1], again[1])


def test_the_shortcut_is_not_taken_when_it_must_not_be():
    from devo_amd.backends import cuda_ba
    sc = _scene(seed=5)
    ii, jj, kk = sc[6:]
    ref = _ba(sc)
    cuda_ba.graph_tables(ii, jj, kk)
    n0 = cuda_ba.import_stats()
    # another tensor with the same values: not the donor's kk
    other = kk.clone()
    a = _ba(sc, kk=other)
    assert cuda_ba.import_stats() == n0 and torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1])
    # the donor's kk again, after a foreign graph went through the same workspace: imported anew, right tables
    perm = torch.randperm(kk.numel(), generator=torch.Generator().manual_seed(1)).to(DEV)
    sc_f = sc[:3] + (sc[3][:, perm], sc[4][:, perm], sc[5], ii[perm], jj[perm], kk[perm])
    f_ref = _ba(sc_f)
    b = _ba(sc)
    assert cuda_ba.import_stats() == n0 + 1 and torch.equal(b[0], ref[0]) and torch.equal(b[1], ref[1])
    f2 = _ba(sc_f)
    assert torch.equal(f2[0], f_ref[0]) and torch.equal(f2[1], f_ref[1])
    # an in-place edit of kk bumps its version: the donor's tables describe the OLD contents
    n1 = cuda_ba.import_stats()
    kk.add_(0)
    c = _ba(sc)
    assert cuda_ba.import_stats() == n1 and torch.equal(c[0], ref[0])
    # switched off
    cuda_ba.graph_tables(ii, jj, kk)
    cuda_ba.SHARE_TABLES = False
    try:
        n2 = cuda_ba.import_stats()
        d = _ba(sc)
        assert cuda_ba.import_stats() == n2 and torch.equal(d[0], ref[0]) and torch.equal(d[1], ref[1])
    finally:
        cuda_ba.SHARE_TABLES = True


def test_patch_ids_beyond_the_patch_buffer_fail_loudly():
    """The operator groups every id below its bound; the BA's own preparation counts ids >= its patch buffer as bad ids (segment 0).  Tables
    with such ids are not imported as they are: the workspace stays unprepared and the adjustment reports it (status -1)."""
    from devo_amd.backends import cuda_ba
    from devo_amd import fastba
    sc = _scene(seed=7)
    ii, jj, kk = sc[6:]
    Np = sc[1].shape[1]
    bad = kk.clone()
    bad[17] = Np + 5
    cuda_ba.graph_tables(ii, jj, bad)
    with pytest.raises(fastba.BAFailure):
        _ba(sc, kk=bad)
    ok = _ba(sc)                                                        # the next healthy call is untouched by it
    assert not torch.equal(ok[0], sc[0])


def test_the_update_operator_offers_its_tables():
    """Update.forward on a new graph (devo.py:311) leaves the donor behind: the BA of the same frame imports."""
    from devo_amd.update import Update
    from devo_amd.backends import cuda_ba
    sc = _scene(seed=3)
    ii, jj, kk = sc[6:]
    E = kk.numel()
    torch.manual_seed(0)
    upd = Update(3).to(DEV).half().eval()
    net = torch.zeros(1, E, 384, device=DEV, dtype=torch.float16)
    inp = torch.randn(1, E, 384, device=DEV, dtype=torch.float16) * 0.1
    corr = torch.randn(1, E, 882, device=DEV, dtype=torch.float16) * 0.1
    ref = _ba(sc)
    n0 = cuda_ba.import_stats()
    with torch.no_grad():
        upd(net, inp, corr, None, ii, jj, kk)
    got = _ba(sc)
    assert cuda_ba.import_stats() == n0 + 1 and torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
