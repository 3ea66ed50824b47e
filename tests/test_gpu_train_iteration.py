"""Two iterations of the reference's training-forward body (devo/enet.py:313-361) — transform, Update, target = centre + delta, two BA steps,
detached state per iteration, a loss on the reprojections, its gradients — composed from THIS package's modules on the GPU, against
tests/golden/train_iter_f64.npz, which tools/gen_golden_train_iter.py produced with the REAL reference modules (devo.enet.Update at width
32, devo.ba.BA, devo.projective_ops) on CPU in fp64.  The lookup is replaced by a seeded random tensor on both sides (the reference has no
CPU lookup): what is pinned is the composition and its adjoint."""
import os
import numpy as np
import pytest
import torch
from util import assert_rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(golden_dir, dt):
    from devo_amd.update import Update
    from devo_amd.ba import BA
    from devo_amd import projective_ops as pops
    from devo_amd.lietorch import SE3
    z = np.load(os.path.join(golden_dir, "train_iter_f64.npz"), allow_pickle=False)
    n, M, H, W, dim, E = (int(z[k]) for k in ("n", "M", "H", "W", "dim", "E"))
    p = 3
    # the generator stream of the fixture's script: edge mask, context features, two corr tensors, two loss weights
    from devo_amd import synth
    g = torch.Generator().manual_seed(int(z["rng_seed"]))
    ii0, _, _ = synth.full_graph(n, M)
    torch.rand(len(ii0), generator=g)
    torch.randn(1, n * M, dim, generator=g, dtype=torch.float64)
    corrs = [torch.randn(1, E, 2 * 49 * p * p, generator=g, dtype=torch.float64) for _ in range(2)]
    lw = [torch.randn(1, E, p, p, 2, generator=g, dtype=torch.float64) for _ in range(2)]
    assert abs(float(sum(c.sum() for c in corrs) + sum(l.abs().sum() for l in lw)) - float(z["rng_checksum"])) < 1e-6, "the generator stream changed"
    t = lambda k: torch.from_numpy(z[k]).to(DEV)
    f = lambda k: t(k).to(dt)
    up = Update(p, dim=dim).to(DEV).to(dt).train()
    up.load_state_dict({k[3:]: f(k) for k in z.files if k.startswith("sd/")})
    ii, jj, kk = t("ii"), t("jj"), t("kk")
    intr, bounds = f("intrinsics"), z["bounds"].tolist()
    imap = f("imap").clone().requires_grad_(True)
    Gs, P = SE3(f("poses").clone()), f("patches").clone()
    net = torch.zeros(1, E, dim, dtype=dt, device=DEV)
    loss = torch.zeros((), dtype=dt, device=DEV)
    seen, kept = {}, []
    for it in range(2):
        Gs = Gs.detach()
        P = P.detach()
        coords = pops.transform(Gs, P, intr, ii, jj, kk)
        net, (delta, weight, _) = up(net, imap[:, kk], corrs[it].to(DEV).to(dt), None, ii, jj, kk)
        delta.retain_grad(); weight.retain_grad()
        kept.append((delta, weight))
        target = coords[..., p // 2, p // 2, :] + delta
        for _ in range(2):
            Gs, P = BA(Gs, P, intr, target, weight, 1e-4, ii, jj, kk, bounds, ep=10, fixedp=1)
        cf = pops.transform(Gs, P, intr, ii, jj, kk)
        loss = loss + (cf * lw[it].to(DEV).to(dt)).sum() * 1e-2 + (Gs.log() ** 2).sum() + (net ** 2).mean()
        seen[f"poses_it{it + 1}"] = Gs.data.detach()
        seen[f"disp_it{it + 1}"] = P.detach()[0, :, 2, 1, 1]
        seen[f"delta_it{it + 1}"] = delta.detach()
        seen[f"weight_it{it + 1}"] = weight.detach()
    loss.backward()
    for it, (dl, wt) in enumerate(kept):
        seen[f"gdelta_it{it + 1}"] = dl.grad
        seen[f"gweight_it{it + 1}"] = wt.grad
    return z, up, imap, loss, seen


@pytest.mark.parametrize("dt,tol,gtol", [(torch.float64, 1e-8, 1e-6), (torch.float32, 2e-5, 3e-4)])      # (measured in fp32: values 1e-6, gradients 3e-5)
def test_two_training_iterations_match_the_reference_modules(golden_dir, dt, tol, gtol):
    z, up, imap, loss, seen = _run(golden_dir, dt)
    for k, v in seen.items():                                            # values; then what arrives at the heads' outputs on the way back
        assert_rel(v.double().cpu(), torch.from_numpy(z[k]), gtol if k.startswith("g") else tol, k)
    assert abs(float(loss.detach()) - float(z["loss"])) <= tol * abs(float(z["loss"]))
    grads = {k: v.grad for k, v in up.named_parameters() if v.grad is not None}
    names = [str(s) for s in z["grad_names"]]
    assert sorted(grads) == names                                           # the same parameters receive a gradient
    norms = torch.tensor([float(grads[k].double().norm()) for k in names], dtype=torch.float64)
    assert_rel(norms, torch.from_numpy(z["grad_norms"]), gtol, "gradient norms of all parameters")
    for k in z.files:
        if k.startswith("grad/"):
            assert_rel(grads[k[5:]].double().cpu(), torch.from_numpy(z[k]), gtol, "d loss / d " + k[5:])
    assert_rel(imap.grad.double().cpu(), torch.from_numpy(z["grad_imap"]), gtol, "d loss / d imap")
