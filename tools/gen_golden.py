#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator; build container only — it imports the reference and oracle/).

Generate tests/golden/*.npz by running the REAL reference Python modules
(devo/projective_ops.py, devo/ba.py, devo/lietorch/groups.py) from /root/reference on CPU.

Runs only in the build container (the GPU box has no /root/reference).  Nothing from the
reference is copied: the outputs written here are data (inputs + expected outputs).

Shims installed before `import devo.*`:
  cuda_ba, cuda_corr   empty modules exposing the attribute names bound at import time
                       (devo/fastba/ba.py:4-5, devo/altcorr/correlation.py:11,28,40,47) — never called here
  torch_scatter        scatter_sum = zeros.index_add  (devo/ba.py:2)
  lietorch_backends    the oracle's SE3 functions (oracle/lie.py `backend`) — so the GROUP VALUES in these
                       goldens come from our own restatement; what the goldens pin is the reference's
                       projective_ops / ba.py / groups.py logic built on top of it.
"""
import os
import sys
import types
import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import lie as olie                      # noqa: E402
from devo_amd import synth                          # noqa: E402


def install_shims():
    for name, attrs in (("cuda_ba", ("forward", "neighbors", "reproject")),
                        ("cuda_corr", ("forward", "backward", "patchify_forward", "patchify_backward"))):
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, None)
        sys.modules[name] = m
    ts = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        shape = list(src.shape)
        shape[dim] = int(dim_size) if dim_size is not None else int(index.max()) + 1
        return torch.zeros(shape, dtype=src.dtype).index_add(dim, index, src)
    ts.scatter_sum = scatter_sum
    sys.modules["torch_scatter"] = ts
    sys.modules["lietorch_backends"] = olie.backend
    sys.path.insert(0, REF)


def scene(n, M, H, W, seed, dtype):
    poses = synth.make_poses(n, seed, dtype=dtype)
    patches, centres = synth.make_patches(n, M, H, W, seed=seed, dtype=dtype)
    intr = synth.make_intrinsics(n, H, W, dtype=dtype)
    ii, jj, kk = synth.full_graph(n, M)
    return poses, patches, intr, ii, jj, kk


def main():
    install_shims()
    from devo import projective_ops as pops
    from devo.ba import BA, CholeskySolver
    from devo.lietorch import SE3

    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    dt = torch.float64
    n, M, H, W = 5, 7, 48, 64
    poses, patches, intr, ii, jj, kk = scene(n, M, H, W, 1234, dt)
    # make the graph less regular: drop ~25% of the edges, shuffle the rest
    g = torch.Generator().manual_seed(77)
    keep = torch.rand(len(ii), generator=g) > 0.25
    perm = torch.randperm(int(keep.sum()), generator=g)
    ii, jj, kk = ii[keep][perm], jj[keep][perm], kk[keep][perm]
    E = len(ii)

    # ---- transform with Jacobians, validity, depth; tonly; flow_mag; point_cloud
    with torch.no_grad():
        c, v, (Ji, Jj, Jz) = pops.transform(SE3(poses), patches, intr, ii, jj, kk, jacobian=True)
        cd = pops.transform(SE3(poses), patches, intr, ii, jj, kk, depth=True)
        ct = pops.transform(SE3(poses), patches, intr, ii, jj, kk, tonly=True)
        fm = pops.flow_mag(SE3(poses), patches, intr, ii, jj, kk, beta=0.5)
        pc = pops.point_cloud(SE3(poses), patches, intr, torch.arange(n * M) // M)
    np.savez_compressed(os.path.join(outdir, "transform_f64.npz"),
                        poses=poses.numpy(), patches=patches.numpy(), intrinsics=intr.numpy(),
                        ii=ii.numpy(), jj=jj.numpy(), kk=kk.numpy(),
                        coords=c.numpy(), valid=v.numpy(), Ji=Ji.numpy(), Jj=Jj.numpy(), Jz=Jz.numpy(),
                        coords_depth=cd.numpy(), coords_tonly=ct.numpy(), flow_mag=fm.numpy(),
                        point_cloud=pc.numpy(), M=M)

    # ---- BA: ep in {10, 100}, structure_only in {F, T}, two successive calls each
    delta, weight = synth.make_update_outputs(E, 1234, sigma=1.0, dtype=dt)
    with torch.no_grad():
        c0 = pops.transform(SE3(poses), patches, intr, ii, jj, kk)
    target = c0[..., 1, 1, :] + delta
    bounds = [-64, -64, W + 64, H + 64]
    out = dict(poses=poses.numpy(), patches=patches.numpy(), intrinsics=intr.numpy(),
               ii=ii.numpy(), jj=jj.numpy(), kk=kk.numpy(), target=target.numpy(), weight=weight.numpy(),
               bounds=np.array(bounds, dtype=np.float64), lmbda=1e-4)
    for ep in (10.0, 100.0):
        for so in (False, True):
            G, P = SE3(poses.clone()), patches.clone()
            with torch.no_grad():
                for it in range(2):
                    G, P = BA(G, P, intr, target, weight, 1e-4, ii, jj, kk, bounds, ep=ep, fixedp=1,
                              structure_only=so)
                    tag = f"ep{int(ep)}_so{int(so)}_it{it + 1}"
                    out["poses_" + tag] = G.data.numpy().copy()
                    out["patches_" + tag] = P.numpy().copy()
    # lmbda passed as a tensor: ba.py:155-156 reshapes it to C's shape, i.e. it is a PER-PATCH damping [m]
    m_uniq = len(torch.unique(kk))
    lm_t = 1e-4 * (1.0 + torch.arange(m_uniq, dtype=dt))
    out["lmbda_tensor"] = lm_t.numpy()
    with torch.no_grad():
        G, P = BA(SE3(poses.clone()), patches.clone(), intr, target, weight, lm_t,
                  ii, jj, kk, bounds, ep=10.0, fixedp=1)
    out["poses_lmtensor"] = G.data.numpy()
    out["patches_lmtensor"] = P.numpy()

    # ---- gradients through one BA step (the training graph, enet.py:353-356 + train.py loss shape)
    tgt = target.clone().requires_grad_(True)
    wgt = weight.clone().requires_grad_(True)
    G, P = BA(SE3(poses.clone()), patches.clone(), intr, tgt, wgt, 1e-4, ii, jj, kk, bounds, ep=10.0, fixedp=1)
    cf = pops.transform(G, P, intr, ii, jj, kk)
    gw = torch.Generator().manual_seed(5)
    lw = torch.randn(cf.shape, generator=gw, dtype=dt)
    loss = (cf * lw).sum() + (G.log() ** 2).sum()
    loss.backward()
    out.update(loss_weights=lw.numpy(), loss=float(loss), grad_target=tgt.grad.numpy(), grad_weight=wgt.grad.numpy())
    np.savez_compressed(os.path.join(outdir, "ba_train_f64.npz"), **out)

    # ---- CholeskySolver forward/backward on a seeded SPD system
    gs = torch.Generator().manual_seed(9)
    A = torch.randn(1, 12, 12, generator=gs, dtype=dt)
    Hm = (A @ A.transpose(-1, -2) + 12 * torch.eye(12, dtype=dt)).requires_grad_(True)
    b = torch.randn(1, 12, 1, generator=gs, dtype=dt).requires_grad_(True)
    x = CholeskySolver.apply(Hm, b)
    gx = torch.randn(x.shape, generator=gs, dtype=dt)
    x.backward(gx)
    np.savez_compressed(os.path.join(outdir, "cholesky_f64.npz"), H=Hm.detach().numpy(), b=b.detach().numpy(),
                        x=x.detach().numpy(), gx=gx.numpy(), dH=Hm.grad.numpy(), db=b.grad.numpy())

    # ---- groups.py plumbing: retr, matrix, translation, scale, broadcasting act
    X = SE3(poses[:, :, None])                                     # [1,n,1,7]
    gp = torch.Generator().manual_seed(11)
    pts = torch.randn(1, n, 6, 4, generator=gp, dtype=dt)
    a = 0.1 * torch.randn(1, n, 6, generator=gp, dtype=dt)
    np.savez_compressed(os.path.join(outdir, "groups_f64.npz"), poses=poses.numpy(), pts=pts.numpy(), a=a.numpy(),
                        act=(X * pts).numpy(), retr=SE3(poses).retr(a).data.numpy(),
                        matrix=SE3(poses).matrix().numpy(), translation=SE3(poses).translation().numpy(),
                        inv=SE3(poses).inv().data.numpy(), log=SE3(poses).log().numpy(),
                        mul=(SE3(poses) * SE3(poses).inv()[:, [0]]).data.numpy(),
                        scale=SE3(poses).scale(torch.full((1, n), 2.0, dtype=dt)).data.numpy())
    print("golden fixtures written to", outdir)
    for f in sorted(os.listdir(outdir)):
        print(" ", f, os.path.getsize(os.path.join(outdir, f)))


if __name__ == "__main__":
    main()
