"""The two bindings of libdevo_hip.so — devo_amd._C (csrc/bind.cpp, compiled) and devo_amd.backends.* over ctypes (DEVO_BINDING=ctypes) —
hold their own host logic (layout conversion caches, plan hand-over, workspace handling, dtype / contiguity preparation).  Verdict r05
weak 10: drift between them was untested beyond the reference-signature functions' existence.  Here every entry point of the three
modules, the reference's and the package's extended ones, runs on the same seeded inputs once per binding (the switch is read at import:
sub-processes) and the outputs are compared BIT FOR BIT (patchify_backward and corr backward's atomic fallback: to fp32 rounding)."""
import os
import subprocess
import sys
import pytest
import torch

pytestmark = pytest.mark.gpu

_LEG = r"""
import sys, os, torch
sys.path.insert(0, sys.argv[1])
from devo_amd import synth
import devo_amd.backends as B
from devo_amd.backends import cuda_corr, cuda_ba, lietorch_backends as lie
assert (B.native() is None) == (os.environ.get("DEVO_BINDING") == "ctypes")
dev = "cuda"
out = {}
n, M, H, W, C, R = 6, 400, 64, 96, 128, 3
poses = synth.make_poses(n, 5, trans_step=0.01, rot_step=0.002).to(dev)
patches, centres = synth.make_patches(n, M, H, W, seed=5)
patches = patches.to(dev)
intr = synth.make_intrinsics(n, H, W).to(dev)
ii, jj, kk = [t.to(dev) for t in synth.full_graph(n, M)]
fmap, gmap = synth.make_features(n, M, C, H, W, centres, seed=5)
fmap, gmap = fmap.to(dev), gmap.to(dev)
f1 = synth.pyramid_l1(fmap)
E = ii.numel()
delta, weight = [t.to(dev) for t in synth.make_update_outputs(E, 5, sigma=0.3)]
lm = torch.tensor([1e-4], device=dev)
# ---- cuda_ba
coords = cuda_ba.transform(poses, patches, intr, ii, jj, kk, layout="2pp")
out["transform_2pp"] = coords
c_pp2, valid, (Ji, Jj, Jz) = cuda_ba.transform(poses, patches, intr, ii, jj, kk, jacobian=True)
out.update(transform_pp2=c_pp2, valid=valid, Ji=Ji, Jj=Jj, Jz=Jz)
out["reproject"] = cuda_ba.reproject(poses, patches, intr, ii, jj, kk)
ix, jx = cuda_ba.neighbors(kk, jj)
out.update(nb_ix=ix, nb_jx=jx)
target = coords[:, :, :, 1, 1] + delta
P1, Q1 = poses.clone(), patches.clone()
cuda_ba.forward(P1, Q1, intr, target, weight, lm, ii, jj, kk, 1, n, 2)
P1b, Q1b = poses.clone(), patches.clone()
cuda_ba.forward(P1b, Q1b, intr, target, weight, lm, ii, jj, kk, 1, n, 2)           # (the remembered index tables)
out.update(ba_P=P1, ba_Q=Q1, ba_P_again=P1b, ba_Q_again=Q1b)
ws = cuda_ba.workspace(E, patches.shape[1], n - 1, dev)
cuda_ba.prepare(kk, patches.shape[1], n - 1, ws)
nseg, kx, seg, perm = cuda_ba.prepared_tables(ws, E, patches.shape[1], n - 1)
out.update(prep_kx=kx, prep_seg=seg, prep_perm=perm)
P2, Q2 = poses.clone(), patches.clone()
cuda_ba.forward_delta(P2, Q2, intr, coords, delta, weight, lm, ii, jj, kk, 1, n, 2, ws)
out.update(bad_P=P2, bad_Q=Q2)
# ---- cuda_corr: the reference's NCHW tensors, fp32 and fp16; the fused pyramid; plans
for name, dt in (("f32", torch.float32), ("f16", torch.float16)):
    g_, f_, f1_ = gmap.to(dt), fmap.to(dt), f1.to(dt)
    out["corr0_" + name] = cuda_corr.forward(g_, f_, coords, kk, jj, R)[0]
    out["corr1_" + name] = cuda_corr.forward(g_, f1_, coords / 4, kk, jj, R)[0]
    out["pyr_" + name] = cuda_corr.forward_pyramid(g_, [f_, f1_], coords, kk, jj, R, (1, 4))
plan = cuda_corr.plan(coords, jj, n, H, radius=R)
out["plan_sorted"] = torch.sort(plan[:E]).values                 # (the order inside a plan bin is whatever the LDS atomics made it: a plan is a permutation)
out["plan_heavy"] = plan[E : E + 1]
out["pyr_plan"] = cuda_corr.forward_pyramid(gmap, [fmap, f1], coords, kk, jj, R, (1, 4), order=plan)
sub = slice(0, 1500)
grad = torch.randn(1, 1500, 7, 7, 3, 3, generator=torch.Generator().manual_seed(2)).to(dev)
d1, d2 = cuda_corr.backward(gmap, fmap, coords[:, sub], kk[sub], jj[sub], grad, R)
out.update(bwd_d1=d1, bwd_d2=d2)
net = torch.randn(1, 32, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
pc = (torch.rand(1, 50, 2, generator=torch.Generator().manual_seed(4)) * torch.tensor([W - 8.0, H - 8.0]) + 4).to(dev)
pf = cuda_corr.patchify_forward(net, pc, 1)[0]
out["patchify"] = pf
out["patchify_bwd"] = cuda_corr.patchify_backward(net, pc, torch.ones_like(pf), 1)[0]
# ---- lietorch_backends (SE3 = group 3)
g = torch.Generator().manual_seed(6)
a = (torch.randn(64, 6, generator=g) * 0.3).to(dev)
b6 = (torch.randn(64, 6, generator=g) * 0.3).to(dev)
X = lie.expm(3, a); Y = lie.expm(3, b6)
p3 = torch.randn(64, 3, generator=g).to(dev); p4 = torch.randn(64, 4, generator=g).to(dev)
gX = torch.randn(64, 7, generator=g).to(dev); g6 = torch.randn(64, 6, generator=g).to(dev)
out.update(exp=X, log=lie.logm(3, X), inv=lie.inv(3, X), mul=lie.mul(3, X, Y), adj=lie.adj(3, X, b6), adjT=lie.adjT(3, X, b6), act=lie.act(3, X, p3),
           act4=lie.act4(3, X, p4), mat=lie.as_matrix(3, X), jinv=lie.Jinv(3, X, b6))
out["exp_b"] = lie.expm_backward(3, gX, a)[0]; out["log_b"] = lie.logm_backward(3, g6, X)[0]; out["inv_b"] = lie.inv_backward(3, gX, X)[0]
for nm, fn, gr, y in (("mul", lie.mul_backward, gX, Y), ("adj", lie.adj_backward, g6, b6), ("adjT", lie.adjT_backward, g6, b6),
                      ("act", lie.act_backward, torch.randn(64, 3, generator=g).to(dev), p3), ("act4", lie.act4_backward, torch.randn(64, 4, generator=g).to(dev), p4)):
    r = fn(3, gr, X, y)
    out[nm + "_bX"], out[nm + "_by"] = r[0], r[1]
torch.cuda.synchronize()
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[2])
"""


def test_both_bindings_return_the_same_bits(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for binding in ("native", "ctypes"):
        env = dict(os.environ)
        env["DEVO_BINDING"] = binding
        path = str(tmp_path / f"{binding}.pt")
        r = subprocess.run([sys.executable, "-c", _LEG, root, path], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res.append(torch.load(path))
    a, b = res
    assert a.keys() == b.keys() and len(a) >= 45
    loose = {"patchify_bwd", "bwd_d1", "bwd_d2"}                           # fp32 atomics: the order of the adds is not fixed
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert bool(torch.isfinite(a[k].float()).all()), k
        if k in loose:
            assert torch.allclose(a[k], b[k], rtol=1e-4, atol=1e-5 * max(1.0, float(a[k].abs().max()))), k
        else:
            assert torch.equal(a[k], b[k]), f"{k}: the two bindings disagree (max |diff| {float((a[k].double() - b[k].double()).abs().max()):.3e})"
    assert torch.equal(a["ba_P"], a["ba_P_again"]) and torch.equal(a["ba_Q"], a["ba_Q_again"])
