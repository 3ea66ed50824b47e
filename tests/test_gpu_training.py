"""GPU parity of the differentiable (training) path — devo_amd.projective_ops.transform with autograd and
devo_amd.ba.BA over the HIP SE3 ops — against fixtures produced by the REAL reference devo/ba.py and
devo/projective_ops.py (tests/golden/ba_train_f64.npz): values and gradients, fp64 and fp32."""
import os
import numpy as np
import pytest
import torch
from util import assert_rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(golden_dir, dt):
    z = np.load(os.path.join(golden_dir, "ba_train_f64.npz"))
    g = {}
    for k in z.files:
        if z[k].ndim == 0:
            g[k] = z[k].item()
        else:
            t = torch.from_numpy(z[k])
            g[k] = (t.to(dt) if t.is_floating_point() else t).to(DEV)
    return g


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-9), (torch.float32, 1e-4)])
def test_ba_values_golden(golden_dir, dt, tol):
    from devo_amd.ba import BA
    from devo_amd.lietorch import SE3
    g = _load(golden_dir, dt)
    bounds = g["bounds"].tolist()
    for ep in (10.0, 100.0):
        for so in (False, True):
            G, P = SE3(g["poses"].clone()), g["patches"].clone()
            with torch.no_grad():
                for it in range(2):
                    G, P = BA(G, P, g["intrinsics"], g["target"], g["weight"], 1e-4, g["ii"], g["jj"], g["kk"], bounds,
                              ep=ep, fixedp=1, structure_only=so)
                    tag = f"ep{int(ep)}_so{int(so)}_it{it + 1}"
                    assert_rel(G.data, g["poses_" + tag].double(), tol, "poses " + tag)
                    assert_rel(P, g["patches_" + tag].double(), tol, "patches " + tag)


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-7), (torch.float32, 2e-3)])
def test_ba_gradients_golden(golden_dir, dt, tol):
    from devo_amd.ba import BA
    from devo_amd import projective_ops as pops
    from devo_amd.lietorch import SE3
    g = _load(golden_dir, dt)
    tgt = g["target"].clone().requires_grad_(True)
    wgt = g["weight"].clone().requires_grad_(True)
    G, P = BA(SE3(g["poses"].clone()), g["patches"].clone(), g["intrinsics"], tgt, wgt, 1e-4, g["ii"], g["jj"], g["kk"],
              g["bounds"].tolist(), ep=10.0, fixedp=1)
    cf = pops.transform(G, P, g["intrinsics"], g["ii"], g["jj"], g["kk"])
    loss = (cf * g["loss_weights"]).sum() + (G.log() ** 2).sum()
    loss.backward()
    assert abs(float(loss.detach()) - g["loss"]) <= tol * abs(g["loss"])
    assert_rel(tgt.grad, g["grad_target"].double(), tol, "d loss / d target")
    assert_rel(wgt.grad, g["grad_weight"].double(), tol, "d loss / d weight")


def test_ba_with_zero_lambda_keeps_unobserved_patches_finite(golden_dir):
    """lmbda = 0 on a graph that leaves patch slots without edges: 1 / (C + lmbda) must not turn into inf * 0 = NaN for those
    slots (the reference only ever solves for observed patches, ba.py:113-118)."""
    from devo_amd.ba import BA
    from devo_amd.lietorch import SE3
    g = _load(golden_dir, torch.float32)
    keep = g["kk"] != g["kk"][0]                                       # drop every edge of one patch
    G, P = BA(SE3(g["poses"].clone()), g["patches"].clone(), g["intrinsics"], g["target"][:, keep], g["weight"][:, keep], 0.0,
              g["ii"][keep], g["jj"][keep], g["kk"][keep], g["bounds"].tolist(), ep=10.0, fixedp=1)
    assert torch.isfinite(P).all() and torch.isfinite(G.data).all()
    k0 = int(g["kk"][0])
    assert torch.equal(P[0, k0], g["patches"][0, k0])                  # the unobserved patch did not move


def test_training_step_reaches_every_parameter():
    """devo_amd.training: one training step of the update + BA path (configuration 3 at a small size) — finite loss, a gradient
    in every parameter of the reference's bucket (Update operator, both encoders, the scorer), weights move."""
    from devo_amd import training as T
    net, model, opt = T.build_trainer(DEV, 1)
    assert net.num_parameters() == T.N_TOTAL
    batch = T.make_batch("cfg1", 1234, DEV)
    before = torch.cat([q.detach().reshape(-1).clone() for q in net.parameters()])
    opt.zero_grad(set_to_none=True)
    loss = model(batch, iters=3)
    loss.backward()
    assert torch.isfinite(loss)
    missing = [n for n, q in net.named_parameters() if q.grad is None or not torch.isfinite(q.grad).all()]
    assert not missing, missing
    assert all(float(q.grad.abs().max()) > 0 for n, q in net.named_parameters() if (n.startswith("update.") and "d.1" not in n) or n.startswith("patchify."))
    l2 = T.train_step(model, opt, batch, iters=3)
    after = torch.cat([q.detach().reshape(-1) for q in net.parameters()])
    assert torch.isfinite(l2) and not torch.equal(before, after)


@pytest.mark.parametrize("structure_only", [False, True])
def test_fused_solve_matches_the_torch_composition_in_value_and_gradient(golden_dir, structure_only, monkeypatch):
    """devo_amd.ba.BA in fp32 runs the normal equations + Schur + Cholesky (+ their adjoint) as HIP kernels
    (devo_ba_solve_terms / _backward); DEVO_BA_TORCH=1 keeps the torch composition that the fp64 goldens pin to the
    reference.  Same outputs, and the same gradients with respect to EVERY differentiable input: targets, weights, poses
    (through the Jacobians: second-order terms included) and patches — two chained steps, like enet.py:353-356."""
    from devo_amd.ba import BA
    from devo_amd import projective_ops as pops
    from devo_amd.lietorch import SE3
    g = _load(golden_dir, torch.float32)

    def run(torch_path):
        monkeypatch.setenv("DEVO_BA_TORCH", "1" if torch_path else "0")
        tgt = g["target"].clone().requires_grad_(True)
        wgt = g["weight"].clone().requires_grad_(True)
        pos = g["poses"].clone().requires_grad_(True)
        pat = g["patches"].clone().requires_grad_(True)
        G, P = SE3(pos), pat
        for _ in range(2):
            G, P = BA(G, P, g["intrinsics"], tgt, wgt, 1e-4, g["ii"], g["jj"], g["kk"], g["bounds"].tolist(), ep=10.0, fixedp=1,
                      structure_only=structure_only)
        cf = pops.transform(G, P, g["intrinsics"], g["ii"], g["jj"], g["kk"])
        loss = (cf * g["loss_weights"]).sum() + (G.log() ** 2).sum() + (P[:, :, 2] ** 2).sum()
        loss.backward()
        return loss.detach(), G.data.detach(), P.detach(), tgt.grad, wgt.grad, pos.grad, pat.grad

    a, b = run(False), run(True)
    assert abs(float(a[0]) - float(b[0])) <= 1e-4 * abs(float(b[0]))
    for x, y, name, tol in zip(a[1:], b[1:], ("poses", "patches", "d/d target", "d/d weight", "d/d poses", "d/d patches"),
                               (1e-4, 1e-4, 2e-3, 2e-3, 2e-3, 2e-3)):
        assert_rel(x, y, tol, name)


def test_fused_solve_breakdown_returns_zero_pose_update_like_the_reference(golden_dir, monkeypatch):
    """devo/ba.py:16-20: when the Cholesky factorisation breaks down (here: a large negative damping) CholeskySolver returns
    zeros and passes no gradient; the depth update dZ = Q (u - E^T 0) still happens.  The fused HIP step must do the same — its
    workspace is not zero-initialised, so the solver itself has to write the zero update — in value and in gradient."""
    from devo_amd.ba import BA
    from devo_amd.lietorch import SE3
    g = _load(golden_dir, torch.float32)

    def run(torch_path):
        monkeypatch.setenv("DEVO_BA_TORCH", "1" if torch_path else "0")
        tgt = g["target"].clone().requires_grad_(True)
        wgt = g["weight"].clone().requires_grad_(True)
        pat = g["patches"].clone().requires_grad_(True)
        if not torch_path:                                            # poison the caching allocator's free blocks: a fresh workspace is then garbage
            junk = [torch.full((1 << 18,), float("nan"), device=DEV) for _ in range(8)]
            del junk
        G, P = BA(SE3(g["poses"].clone()), pat, g["intrinsics"], tgt, wgt, 1e-4, g["ii"], g["jj"], g["kk"], g["bounds"].tolist(), ep=-1.0e9, fixedp=1)
        (P[:, :, 2] ** 2).sum().backward()
        return G.data.detach(), P.detach(), tgt.grad, wgt.grad, pat.grad

    a, b = run(False), run(True)
    assert torch.equal(a[0], g["poses"]) and torch.equal(b[0], g["poses"])          # no pose moved
    assert bool(torch.isfinite(a[1]).all()) and float((a[1] - g["patches"]).abs().max()) > 0     # the depths did
    for x, y, name in zip(a[1:], b[1:], ("patches", "d/d target", "d/d weight", "d/d patches")):
        assert bool(torch.isfinite(x).all()), name
        assert_rel(x, y, 2e-3, name)


@pytest.mark.parametrize("mode", ["coords", "depth", "tonly", "jacobian"])
def test_fused_transform_adjoint_matches_the_autograd_composition(golden_dir, mode, monkeypatch):
    """projective_ops.transform with gradients: ONE forward kernel + ONE adjoint kernel (devo_transform_vjp, the same
    arithmetic on dual numbers) against the composition over the SE3 ops that the fp64 goldens pin to the reference
    (DEVO_TRANSFORM_TORCH=1) — outputs and the gradients with respect to poses (lietorch's 6-of-7 convention) and patches,
    through the coordinates and, for jacobian=True, through Ji / Jj / Jz (second-order terms)."""
    from devo_amd import projective_ops as pops
    from devo_amd.lietorch import SE3
    g = _load(golden_dir, torch.float32)
    gen = torch.Generator().manual_seed(5)

    def run(torch_path):
        monkeypatch.setenv("DEVO_TRANSFORM_TORCH", "1" if torch_path else "0")
        pos = g["poses"].clone().requires_grad_(True)
        pat = g["patches"].clone().requires_grad_(True)
        kw = dict(depth=(mode == "depth"), tonly=(mode == "tonly"), jacobian=(mode == "jacobian"))
        out = pops.transform(SE3(pos), pat, g["intrinsics"], g["ii"], g["jj"], g["kk"], **kw)
        outs = [out] if not isinstance(out, tuple) else [out[0], *out[2]]
        loss = 0.0
        gen.manual_seed(5)
        for o in outs:
            loss = loss + (o * torch.randn(o.shape, generator=gen).to(o.device)).sum()
        loss.backward()
        return [o.detach() for o in outs], pos.grad, pat.grad

    (oa, pa, qa), (ob, pb, qb) = run(False), run(True)
    for x, y in zip(oa, ob):
        assert_rel(x, y, 1e-5, f"{mode}: output")
    assert float(pa[..., 6].abs().max()) == 0.0                       # the 7th slot of a group gradient stays empty
    assert_rel(pa, pb, 2e-4, f"{mode}: d/d poses")
    assert_rel(qa, qb, 2e-4, f"{mode}: d/d patches")


def _dp_rank(outdir):
    import json, os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import torch.distributed as dist
    from devo_amd import distributed as D, training as T
    rank, world = D.init_from_env("gloo", "cuda:0")              # two ranks share the one GPU of the test box: gloo instead of RCCL
    torch.cuda.set_device(0)
    net, model, opt = T.build_trainer("cuda:0", world)
    batch = T.make_batch("cfg1", 1234 + rank, "cuda:0")         # every rank its own sequence (train.py:91-93)
    opt.zero_grad(set_to_none=True)
    loss = model(batch, iters=2)
    loss.backward()                                              # DDP all-reduces (mean) the 3 397 061-element bucket
    g = torch.cat([q.grad.reshape(-1) for q in net.parameters()]).double()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
    opt.step()
    w = torch.cat([q.detach().reshape(-1) for q in net.parameters()]).double()
    with open(os.path.join(outdir, f"dp{rank}.json"), "w") as f:
        json.dump([rank, float(loss), float(g.sum()), float(g.abs().sum()), float(w.sum()), int(g.numel()), bool(torch.isfinite(g).all())], f)
    dist.destroy_process_group()


def test_data_parallel_training_step_on_the_gpu(tmp_path):
    """BASELINE configuration 4's step through DistributedDataParallel with the REAL forward (encoders, lookup, Update, differentiable
    BA) — two ranks through the repo's launcher, different sequences, one gradient bucket: after backward both ranks hold the same
    (averaged) gradient of all 3 397 061 parameters and, after the optimiser step, the same weights.  (One GPU here: both ranks use
    it and gloo carries the all-reduce; on the multi-GPU node the same code runs over RCCL.)"""
    import json
    from devo_amd import distributed as D
    D.launch(_dp_rank, 2, (str(tmp_path),))
    a, b = (json.load(open(tmp_path / f"dp{r}.json")) for r in range(2))
    assert a[5] == b[5] == 3_397_061 and a[6] and b[6]
    assert a[1] != b[1]                                        # different sequences, different losses
    assert abs(a[2] - b[2]) <= 1e-9 * max(1.0, a[3]) and abs(a[3] - b[3]) <= 1e-9 * a[3]     # the same reduced gradient
    assert abs(a[4] - b[4]) <= 1e-9 * abs(a[4])              # the same weights after the step
