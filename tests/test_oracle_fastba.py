"""Known-answer / cross-variant tests that pin the fastba oracle (no reference tests or goldens exist
for cuda_ba: SURVEY.md §4, §8c).  The strongest pin: on problems where the two reference BA variants
coincide (ep=1 damping, no gate/clamp active) the restatement of the CUDA BA
must reproduce the restatement of devo/ba.py, which is itself pinned against the real reference
Python (tests/test_oracle_golden.py)."""
import torch
from oracle import fastba as F
from oracle import pops
from oracle import se3 as K
from oracle.lie import SE3
from devo_amd import synth

DT = torch.float64


def scene(n=5, M=6, H=48, W=64, seed=3, sigma=0.5):
    poses = synth.make_poses(n, seed, dtype=DT)
    patches, _ = synth.make_patches(n, M, H, W, seed=seed, dtype=DT)
    intr = synth.make_intrinsics(n, H, W, dtype=DT)
    ii, jj, kk = synth.full_graph(n, M)
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(len(ii), generator=g)[: int(0.8 * len(ii))]
    ii, jj, kk = ii[perm], jj[perm], kk[perm]
    delta, weight = synth.make_update_outputs(len(ii), seed, sigma=sigma, dtype=DT)
    c0 = pops.transform(SE3(poses), patches, intr, ii, jj, kk)
    target = c0[..., 1, 1, :] + delta
    return poses, patches, intr, target, weight, ii, jj, kk


def test_reproject_identity_and_vs_transform():
    poses, patches, intr, target, weight, ii, jj, kk = scene()
    same = F.reproject(poses, patches, intr, ii, ii, kk, dtype=DT)
    assert torch.allclose(same[0], patches[0, kk, :2], atol=1e-10)        # identity relative pose -> patch grid
    c = pops.transform(SE3(poses), patches, intr, ii, jj, kk)             # [1,E,P,P,2], Z>0.1 here so no clamp
    r = F.reproject(poses, patches, intr, ii, jj, kk, dtype=DT)           # [1,E,2,P,P]
    assert torch.allclose(r, c.permute(0, 1, 4, 2, 3), atol=1e-9)


def test_jacobians_match_python_variant_and_fd():
    poses, patches, intr, target, weight, ii, jj, kk = scene()
    J = F.residuals_and_jacobians(poses[0], patches[0], intr[0, 0], target[0], weight[0], ii, jj, kk)
    _, _, (Ji, Jj, Jz) = pops.transform(SE3(poses), patches, intr, ii, jj, kk, jacobian=True)
    assert torch.allclose(J["Jj"], Jj[0], atol=1e-9)
    assert torch.allclose(J["Ji"], -Ji[0], atol=1e-9)                     # sign folded into the accumulation (ba_cuda.cu:294-322)
    assert torch.allclose(J["Jz"], Jz[0, :, :, 0], atol=1e-9)
    # finite differences of the projected centre wrt left perturbation of pose j / pose i / inverse depth
    h = 1e-6

    def centre(p, q):
        return F.reproject(p, q, intr, ii, jj, kk, dtype=DT)[0, :, :, 1, 1]
    e = 7
    for k in range(6):
        d = torch.zeros(1, poses.shape[1], 6, dtype=DT)
        d[0, jj[e], k] = h
        pp = K.mul(K.expm(d[0]), poses[0])[None]
        pm = K.mul(K.expm(-d[0]), poses[0])[None]
        fd = (centre(pp, patches)[e] - centre(pm, patches)[e]) / (2 * h)
        if ii[e] != jj[e]:
            assert torch.allclose(fd, J["Jj"][e, :, k], atol=1e-5)
        d = torch.zeros(1, poses.shape[1], 6, dtype=DT)
        d[0, ii[e], k] = h
        pp = K.mul(K.expm(d[0]), poses[0])[None]
        pm = K.mul(K.expm(-d[0]), poses[0])[None]
        fd = (centre(pp, patches)[e] - centre(pm, patches)[e]) / (2 * h)
        if ii[e] != jj[e]:
            assert torch.allclose(fd, -J["Ji"][e, :, k], atol=1e-5)
    qp, qm = patches.clone(), patches.clone()
    qp[0, kk[e], 2] += h
    qm[0, kk[e], 2] -= h
    fd = (centre(poses, qp)[e] - centre(poses, qm)[e]) / (2 * h)
    assert torch.allclose(fd, J["Jz"][e], atol=1e-5)


def test_zero_residual_gives_zero_update():
    poses, patches, intr, target, weight, ii, jj, kk = scene(sigma=0.0)
    p2, q2 = F.ba(poses, patches, intr, target, weight, torch.tensor([1e-4]), ii, jj, kk, 1, poses.shape[1], 2, dtype=DT)
    assert torch.allclose(p2, poses, atol=1e-9)
    assert torch.allclose(q2, patches, atol=1e-9)


def test_cuda_variant_equals_python_variant_when_they_coincide():
    poses, patches, intr, target, weight, ii, jj, kk = scene()
    n = poses.shape[1]
    for t0 in (1, 2):
        p2, q2 = F.ba(poses, patches, intr, target, weight, torch.tensor([1e-4]), ii, jj, kk, t0, n, 1, dtype=DT)
        G, P = pops.BA(SE3(poses.clone()), patches.clone(), intr, target, weight, 1e-4, ii, jj, kk,
                       [-64, -64, 2 * float(intr[0, 0, 2]) + 64, 2 * float(intr[0, 0, 3]) + 64], ep=1.0, fixedp=t0)
        # quaternions: the CUDA retraction does not renormalise; compare as rotations
        assert torch.allclose(p2[..., :3], G.data[..., :3], atol=1e-8)
        qa = p2[..., 3:] / p2[..., 3:].norm(dim=-1, keepdim=True)
        assert torch.allclose(qa, G.data[..., 3:], atol=1e-8)
        free = (P[0, :, 2, 0, 0] > 1.001e-3) & (P[0, :, 2, 0, 0] < 9.99)    # depth clamps differ (Appendix B)
        assert int(free.sum()) > 0.9 * free.numel()
        assert torch.allclose(q2[0, free], P[0, free], atol=1e-8)
    # two iterations inside one call == two python calls
    p2, q2 = F.ba(poses, patches, intr, target, weight, torch.tensor([1e-4]), ii, jj, kk, 1, n, 2, dtype=DT)
    G, P = SE3(poses.clone()), patches.clone()
    for _ in range(2):
        G, P = pops.BA(G, P, intr, target, weight, 1e-4, ii, jj, kk, [-64, -64, 1e9, 1e9], ep=1.0, fixedp=1)
    assert torch.allclose(p2[..., :3], G.data[..., :3], atol=1e-7)
    assert torch.allclose(q2, P, atol=1e-7)


def test_structure_only_and_depth_rules():
    poses, patches, intr, target, weight, ii, jj, kk = scene()
    n = poses.shape[1]
    p2, q2 = F.ba(poses, patches, intr, target, weight, torch.tensor([1e-4]), ii, jj, kk, n, n, 1, dtype=DT)
    assert torch.equal(p2, poses)                                        # t1 - t0 == 0: poses untouched (ba_cuda.cu:494-506)
    assert not torch.allclose(q2, patches)
    assert torch.equal(q2[:, :, :2], patches[:, :, :2])
    d = q2[0, :, 2]
    assert torch.all(d == d[:, :1, :1]) and float(d.min()) >= 1e-4 and float(d.max()) <= 20
    # unobserved patches keep their depth
    seen = torch.zeros(patches.shape[1], dtype=torch.bool)
    seen[kk] = True
    assert torch.equal(q2[0, ~seen], patches[0, ~seen])
    # fp32 run close to fp64 run
    p32, q32 = F.ba(poses.float(), patches.float(), intr.float(), target.float(), weight.float(), torch.tensor([1e-4]),
                    ii, jj, kk, 1, n, 2, dtype=torch.float32)
    p64, q64 = F.ba(poses, patches, intr, target, weight, torch.tensor([1e-4]), ii, jj, kk, 1, n, 2, dtype=DT)
    assert torch.allclose(p32.double(), p64, atol=2e-4) and torch.allclose(q32.double(), q64, atol=2e-4)


def test_neighbors_kat():
    #            e: 0  1  2  3  4  5  6
    ii = torch.tensor([5, 5, 9, 5, 9, 5, 2])
    jj = torch.tensor([3, 1, 4, 3, 0, 2, 7])
    ix, jx = F.neighbors(ii, jj)
    # patch 5: edges sorted by jj (stable): e1(j1), e5(j2), e0(j3), e3(j3)
    assert ix.tolist() == [5, -1, 4, 0, -1, 1, -1]
    assert jx.tolist() == [3, 5, -1, -1, 2, 0, -1]


def test_unique_kk_kat():
    kk = torch.tensor([7, 3, 7, 10, 3])
    kx, ku = F.unique_kk(kk)
    assert kx.tolist() == [3, 7, 10] and ku.tolist() == [1, 0, 1, 2, 0]
