"""Pins the oracle's SE3 maths with the algebraic identities the reference's own lietorch test
script uses (devo/lietorch/run_tests.py:16-52, fp64, atol 1e-8) and with fp64 finite differences
of every backward rule (run_tests.py:56-226 check the same Jacobians numerically)."""
import torch
from oracle import se3 as K

torch.manual_seed(0)
DT = torch.float64


def rnd(*s):
    return torch.randn(*s, dtype=DT)


def test_exp_log_roundtrip():
    a = 0.2 * rnd(500, 6)
    assert torch.allclose(K.logm(K.expm(a)), a, atol=1e-8)


def test_exp_small_angle_branch():
    a = torch.cat([rnd(64, 3), 1e-8 * rnd(64, 3)], -1)
    X = K.expm(a)
    assert torch.allclose(X[:, :3], a[:, :3], atol=1e-7)
    assert torch.allclose(K.logm(X), a, atol=1e-8)


def test_inverse():
    X = K.expm(rnd(300, 6))
    I = K.mul(X, K.inv(X))
    assert torch.allclose(K.logm(I), torch.zeros(300, 6, dtype=DT), atol=1e-8)


def test_adjoint_identity():
    X, a = K.expm(rnd(200, 6)), rnd(200, 6)
    Y1 = K.mul(X, K.expm(a))
    Y2 = K.mul(K.expm(K.adj(X, a)), X)
    assert torch.allclose(K.logm(K.mul(Y1, K.inv(Y2))), torch.zeros(200, 6, dtype=DT), atol=1e-8)


def test_adjT_is_transpose():
    X, a, b = K.expm(rnd(50, 6)), rnd(50, 6), rnd(50, 6)
    assert torch.allclose((K.adj(X, a) * b).sum(-1), (a * K.adjT(X, b)).sum(-1), atol=1e-10)


def test_act_matches_matrix():
    X, p = K.expm(rnd(100, 6)), rnd(100, 4)
    T = K.as_matrix(X)
    assert torch.allclose(K.act4(X, p), (T @ p[..., None])[..., 0], atol=1e-10)
    assert torch.allclose(K.act(X, p[:, :3]), (T[:, :3, :3] @ p[:, :3, None])[..., 0] + T[:, :3, 3], atol=1e-10)


def test_quaternion_renormalised_on_load():
    X = K.expm(rnd(20, 6))
    Xs = X.clone()
    Xs[:, 3:] *= 3.0
    p = rnd(20, 4)
    assert torch.allclose(K.act4(X, p), K.act4(Xs, p), atol=1e-12)
    assert torch.allclose(K.inv(X), K.inv(Xs), atol=1e-12)


def _numjac_left(f, X, extra, h=1e-6):
    """d f(Exp(d) X) / d d  at d = 0 (central differences) -> [batch, out, 6]"""
    cols = []
    for k in range(6):
        d = torch.zeros(X.shape[0], 6, dtype=DT)
        d[:, k] = h
        fp = f(K.mul(K.expm(d), X), *extra)
        fm = f(K.mul(K.expm(-d), X), *extra)
        cols.append((fp - fm) / (2 * h))
    return torch.stack(cols, -1)


def _group_diff(Yp, Ym, h):
    """tangent of the output curve: Log(Yp Ym^-1) / 2h"""
    return K.logm(K.mul(Yp, K.inv(Ym))) / (2 * h)


def test_backward_rules_fd():
    B = 40
    X, Y = K.expm(0.5 * rnd(B, 6)), K.expm(0.5 * rnd(B, 6))
    a, p, g6, g4 = rnd(B, 6), rnd(B, 4), rnd(B, 6), rnd(B, 4)
    h = 1e-6
    # act4: true gradient of <g4, act4(X, p)> wrt left perturbation of X and wrt p
    Jx = _numjac_left(K.act4, X, (p,))
    dX, dp = K.act4_backward(g4, X, p)
    assert torch.allclose(dX[:, :6], (g4[:, None] @ Jx)[:, 0], atol=1e-6)
    assert torch.allclose(dp, (g4[:, None] @ K.as_matrix(X))[:, 0], atol=1e-10)
    # act3
    Jx = _numjac_left(K.act, X, (p[:, :3],))
    dX, dp3 = K.act_backward(g4[:, :3], X, p[:, :3])
    assert torch.allclose(dX[:, :6], (g4[:, None, :3] @ Jx)[:, 0], atol=1e-6)
    # adjT / adj wrt X and a
    for fwd, bwd in ((K.adjT, K.adjT_backward), (K.adj, K.adj_backward)):
        Jx = _numjac_left(fwd, X, (a,))
        dX, da = bwd(g6, X, a)
        assert torch.allclose(dX[:, :6], (g6[:, None] @ Jx)[:, 0], atol=1e-6)
        cols = []
        for k in range(6):
            e = torch.zeros(B, 6, dtype=DT); e[:, k] = 1
            cols.append(fwd(X, e))
        Ja = torch.stack(cols, -1)
        assert torch.allclose(da, (g6[:, None] @ Ja)[:, 0], atol=1e-10)
    # inv: output tangent = d Log(inv(Exp(d)X) inv(X)^-1)
    cols = []
    for k in range(6):
        d = torch.zeros(B, 6, dtype=DT); d[:, k] = h
        cols.append(_group_diff(K.inv(K.mul(K.expm(d), X)), K.inv(K.mul(K.expm(-d), X)), h))
    J = torch.stack(cols, -1)
    assert torch.allclose(K.inv_backward(torch.cat([g6, g6[:, :1]], -1), X)[:, :6], (g6[:, None] @ J)[:, 0], atol=1e-6)
    # mul wrt Y
    cols = []
    for k in range(6):
        d = torch.zeros(B, 6, dtype=DT); d[:, k] = h
        cols.append(_group_diff(K.mul(X, K.mul(K.expm(d), Y)), K.mul(X, K.mul(K.expm(-d), Y)), h))
    J = torch.stack(cols, -1)
    dXm, dYm = K.mul_backward(torch.cat([g6, g6[:, :1]], -1), X, Y)
    assert torch.allclose(dYm[:, :6], (g6[:, None] @ J)[:, 0], atol=1e-6)
    assert torch.allclose(dXm[:, :6], g6, atol=0)
    # exp: output tangent wrt a
    cols = []
    for k in range(6):
        d = torch.zeros(B, 6, dtype=DT); d[:, k] = h
        cols.append(_group_diff(K.expm(a + d), K.expm(a - d), h))
    J = torch.stack(cols, -1)
    assert torch.allclose(K.expm_backward(torch.cat([g6, g6[:, :1]], -1), a), (g6[:, None] @ J)[:, 0], atol=1e-6)
    # log: d Log(Exp(d) X)
    Jx = _numjac_left(K.logm, X, ())
    assert torch.allclose(K.logm_backward(g6, X)[:, :6], (g6[:, None] @ Jx)[:, 0], atol=1e-6)


def test_left_jacobian_inverse_consistency():
    a = 0.7 * rnd(50, 6)
    I = K.left_jacobian(a) @ K.left_jacobian_inverse(a)
    assert torch.allclose(I, torch.eye(6, dtype=DT).expand(50, 6, 6), atol=1e-9)
