"""GPU parity of the SE3 kernels (lietorch_backends.*) against the CPU oracle (oracle/se3.py), forward and
backward, fp64 (tight) and fp32; plus the reference's own algebraic identities (run_tests.py:16-52) through
the devo_amd.lietorch.SE3 type, and autograd through the HIP ops."""
import pytest
import torch
from oracle import se3 as K

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(n, seed=0, dt=torch.float64):
    g = torch.Generator().manual_seed(seed)
    a = 0.5 * torch.randn(n, 6, generator=g, dtype=torch.float64)
    a[: n // 8, 3:] *= 1e-9                                  # exercise the small-angle branches
    X = K.expm(a)
    X[n // 2:, 3:] *= 1.7                                    # un-normalised quaternions must be renormalised on load
    Y = K.expm(0.5 * torch.randn(n, 6, generator=g, dtype=torch.float64))
    b = torch.randn(n, 6, generator=g, dtype=torch.float64)
    p4 = torch.randn(n, 4, generator=g, dtype=torch.float64)
    g7 = torch.randn(n, 7, generator=g, dtype=torch.float64)
    return [t.to(dt) for t in (a, X, Y, b, p4, g7)]


@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-11), (torch.float32, 2e-5)])
def test_forward_backward_vs_oracle(dt, tol):
    from devo_amd.backends import lietorch_backends as B
    a, X, Y, b, p4, g7 = _rand(1000, 1, dt)
    d = lambda t: t.to(DEV).contiguous()
    o = lambda t: t.double()
    cases = [
        ("expm", B.expm(3, d(a)), K.expm(o(a))),
        ("logm", B.logm(3, d(X)), K.logm(o(X))),
        ("inv", B.inv(3, d(X)), K.inv(o(X))),
        ("mul", B.mul(3, d(X), d(Y)), K.mul(o(X), o(Y))),
        ("adj", B.adj(3, d(X), d(b)), K.adj(o(X), o(b))),
        ("adjT", B.adjT(3, d(X), d(b)), K.adjT(o(X), o(b))),
        ("act", B.act(3, d(X), d(p4[:, :3].contiguous())), K.act(o(X), o(p4[:, :3]))),
        ("act4", B.act4(3, d(X), d(p4)), K.act4(o(X), o(p4))),
        ("as_matrix", B.as_matrix(3, d(X)), K.as_matrix(o(X))),
        ("Jinv", B.Jinv(3, d(X), d(b)), K.jinv(o(X), o(b))),
        ("expm_backward", B.expm_backward(3, d(g7), d(a))[0], K.expm_backward(o(g7), o(a))),
        ("logm_backward", B.logm_backward(3, d(b), d(X))[0], K.logm_backward(o(b), o(X))),
        ("inv_backward", B.inv_backward(3, d(g7), d(X))[0], K.inv_backward(o(g7), o(X))),
    ]
    for name, got, ref in cases:
        err = (got.double().cpu() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), f"{name}: {err}"
    pairs = [
        ("mul_backward", B.mul_backward(3, d(g7), d(X), d(Y)), K.mul_backward(o(g7), o(X), o(Y))),
        ("adj_backward", B.adj_backward(3, d(b), d(X), d(a)), K.adj_backward(o(b), o(X), o(a))),
        ("adjT_backward", B.adjT_backward(3, d(b), d(X), d(a)), K.adjT_backward(o(b), o(X), o(a))),
        ("act_backward", B.act_backward(3, d(p4[:, :3].contiguous()), d(X), d(g7[:, :3].contiguous())),
         K.act_backward(o(p4[:, :3]), o(X), o(g7[:, :3]))),
        ("act4_backward", B.act4_backward(3, d(p4), d(X), d(g7[:, :4].contiguous())),
         K.act4_backward(o(p4), o(X), o(g7[:, :4]))),
    ]
    for name, got, ref in pairs:
        for gt, rf in zip(got, ref):
            err = (gt.double().cpu() - rf).abs().max().item()
            assert err <= tol * max(1.0, rf.abs().max().item()), f"{name}: {err}"


def test_reference_identities_fp64():
    """devo/lietorch/run_tests.py:16-52 on the GPU, float64, atol 1e-8."""
    from devo_amd.lietorch import SE3
    torch.manual_seed(0)
    a = 0.2 * torch.randn(2, 3, 4, 5, 6, device=DEV, dtype=torch.float64)
    assert torch.allclose(SE3.exp(a).log(), a, atol=1e-8)
    X = SE3.exp(0.1 * torch.randn(2, 3, 4, 5, 6, device=DEV, dtype=torch.float64))
    assert torch.allclose((X * X.inv()).log(), torch.zeros_like(a[..., :6]), atol=1e-8)
    X = SE3.exp(torch.randn(2, 3, 4, 5, 6, device=DEV, dtype=torch.float64))
    b = torch.randn(2, 3, 4, 5, 6, device=DEV, dtype=torch.float64)
    Y1, Y2 = X * SE3.exp(b), SE3.exp(X.adj(b)) * X
    assert torch.allclose((Y1 * Y2.inv()).log(), torch.zeros_like(b), atol=1e-8)
    X = SE3.exp(torch.randn(1, 6, device=DEV, dtype=torch.float64))
    p = torch.randn(1, 3, device=DEV, dtype=torch.float64)
    ph = torch.cat([p, torch.ones_like(p[..., :1])], -1)
    assert torch.allclose(X.act(p), (X.matrix() @ ph[..., None])[..., 0][..., :3], atol=1e-8)


def test_groups_golden_on_gpu(golden_dir):
    import os
    import numpy as np
    from devo_amd.lietorch import SE3
    z = np.load(os.path.join(golden_dir, "groups_f64.npz"))
    g = {k: torch.from_numpy(z[k]).to(DEV) for k in z.files}
    X = SE3(g["poses"])
    assert torch.allclose(SE3(g["poses"][:, :, None]) * g["pts"], g["act"], atol=1e-11)
    assert torch.allclose(X.retr(g["a"]).data, g["retr"], atol=1e-11)
    assert torch.allclose(X.matrix(), g["matrix"], atol=1e-11)
    assert torch.allclose(X.translation(), g["translation"], atol=1e-11)
    assert torch.allclose(X.inv().data, g["inv"], atol=1e-11)
    assert torch.allclose(X.log(), g["log"], atol=1e-11)
    assert torch.allclose((X * X.inv()[:, [0]]).data, g["mul"], atol=1e-11)
    assert torch.allclose(X.scale(torch.full((1, X.shape[1]), 2.0, device=DEV, dtype=torch.float64)).data, g["scale"], atol=1e-12)


def test_autograd_matches_finite_differences():
    """End-to-end real-valued function through Exp, Mul, Inv, Act4, AdjT, Log on the GPU (fp64)."""
    from devo_amd.lietorch import SE3
    torch.manual_seed(3)
    a = (0.3 * torch.randn(7, 6, device=DEV, dtype=torch.float64)).requires_grad_(True)
    Y = SE3.exp(0.4 * torch.randn(7, 6, device=DEV, dtype=torch.float64))
    p = torch.randn(7, 4, device=DEV, dtype=torch.float64)
    b = torch.randn(7, 6, device=DEV, dtype=torch.float64)

    def f(a_):
        X = SE3.exp(a_)
        G = (Y * X.inv()) * X.retr(0.1 * a_)
        return (G.act(p) ** 2).sum() + (G.adjT(b) * b).sum() + (G.log() ** 2).sum()

    f(a).backward()
    num = torch.zeros_like(a)
    h = 1e-6
    with torch.no_grad():
        for i in range(7):
            for k in range(6):
                d = torch.zeros_like(a)
                d[i, k] = h
                num[i, k] = (f(a + d) - f(a - d)) / (2 * h)
    assert torch.allclose(a.grad, num, rtol=1e-5, atol=1e-6)
