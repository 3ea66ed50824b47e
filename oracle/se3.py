"""ORACLE (test infrastructure, never shipped / never on the product path).

CPU restatement, in plain torch-CPU tensor arithmetic, of the SE(3) subset of the
reference's `lietorch_backends` extension (reference: devo/lietorch/include/se3.h,
so3.h, common.h; kernels devo/lietorch/src/lietorch_gpu.cu:20-294).  The reference
needs Eigen 3.4.0 (not vendored, setup.py:29) and a CUDA device, so it cannot be
built or run in this container: parity for the group values is pinned by the
algebraic identities of devo/lietorch/run_tests.py:16-226 plus fp64 finite
differences (tests/test_oracle_se3.py), not by reference output ("parity unpinned").

Element layout: X[..., 7] = (tx, ty, tz, qx, qy, qz, qw); tangent a[..., 6] = (tau, phi).
Every function works on [batch, dim] tensors and keeps the input dtype.
Backward functions follow the lietorch convention: the gradient of a group element
is a 6-vector (left-perturbation tangent) stored in the first 6 of 7 slots.
"""
import torch

EPS = 1e-6  # common.h:7


# --------------------------------------------------------------------------- helpers
def _hat(v):
    """so3.h:104-112  [v]x"""
    x, y, z = v.unbind(-1)
    o = torch.zeros_like(x)
    return torch.stack([o, -z, y, z, o, -x, -y, x, o], -1).view(*v.shape[:-1], 3, 3)


def _qnormalize(q):
    """so3.h:31-37: every load of a quaternion re-normalises it."""
    return q / q.norm(dim=-1, keepdim=True)


def _qmul(a, b):
    """Eigen quaternion product, coefficient order (x, y, z, w)."""
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz], -1)


def _qconj(q):
    return torch.cat([-q[..., :3], q[..., 3:]], -1)


def _qrot(q, p):
    """so3.h:51-56  p + w*uv + qv x uv, uv = 2 qv x p"""
    qv, w = q[..., :3], q[..., 3:]
    uv = torch.linalg.cross(qv, p, dim=-1)
    uv = uv + uv
    return p + w * uv + torch.linalg.cross(qv, uv, dim=-1)


def _qmat(q):
    """Eigen toRotationMatrix of a unit quaternion."""
    x, y, z, w = q.unbind(-1)
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.ones_like(x)
    return torch.stack([
        one - (tyy + tzz), txy - twz, txz + twy,
        txy + twz, one - (txx + tzz), tyz - twx,
        txz - twy, tyz + twx, one - (txx + tyy)], -1).view(*q.shape[:-1], 3, 3)


def _split(X):
    return X[..., :3], _qnormalize(X[..., 3:7])


# --------------------------------------------------------------------------- SO3 pieces
def so3_exp(phi):
    """so3.h:153-168"""
    theta2 = (phi * phi).sum(-1, keepdim=True)
    theta = theta2.sqrt()
    theta4 = theta2 * theta2
    small = theta < EPS
    ts = torch.where(small, torch.ones_like(theta), theta)
    imag = torch.where(small, 0.5 - (1.0 / 48.0) * theta2 + (1.0 / 3840.0) * theta4,
                       torch.sin(0.5 * ts) / ts)
    real = torch.where(small, 1.0 - (1.0 / 8.0) * theta2 + (1.0 / 384.0) * theta4,
                       torch.cos(0.5 * ts))
    q = torch.cat([imag * phi, real], -1)
    return _qnormalize(q)  # SO3(q) constructor normalises (so3.h:27-29)


def so3_log(q):
    """so3.h:115-151 (atan-based log)"""
    qv, w = q[..., :3], q[..., 3:]
    squared_n = (qv * qv).sum(-1, keepdim=True)
    n = squared_n.sqrt()
    small = squared_n < EPS * EPS
    ws = torch.where(w.abs() < EPS, torch.ones_like(w), w)
    ns = torch.where(small, torch.ones_like(n), n)
    a_small = 2.0 / ws - (2.0 / 3.0) * squared_n / (ws * ws * ws)
    a_w0 = torch.where(w > 0, torch.pi / ns, -torch.pi / ns)
    a_gen = 2.0 * torch.atan(ns / ws) / ns
    fac = torch.where(small, a_small, torch.where(w.abs() < EPS, a_w0, a_gen))
    return fac * qv


def so3_left_jacobian(phi):
    """so3.h:170-187"""
    Phi = _hat(phi)
    Phi2 = Phi @ Phi
    theta2 = (phi * phi).sum(-1)[..., None, None]
    theta = theta2.sqrt()
    small = theta < EPS
    t2 = torch.where(small, torch.ones_like(theta2), theta2)
    t = torch.where(small, torch.ones_like(theta), theta)
    c1 = torch.where(small, 0.5 - (1.0 / 24.0) * theta2, (1.0 - torch.cos(t)) / t2)
    c2 = torch.where(small, 1.0 / 6.0 - (1.0 / 120.0) * theta2, (t - torch.sin(t)) / (t2 * t))
    I = torch.eye(3, dtype=phi.dtype).expand_as(Phi)
    return I + c1 * Phi + c2 * Phi2


def so3_left_jacobian_inverse(phi):
    """so3.h:189-205"""
    Phi = _hat(phi)
    Phi2 = Phi @ Phi
    theta2 = (phi * phi).sum(-1)[..., None, None]
    theta = theta2.sqrt()
    small = theta < EPS
    t = torch.where(small, torch.ones_like(theta), theta)
    half = 0.5 * t
    c2 = torch.where(small, torch.full_like(theta, 1.0 / 12.0),
                     (1.0 - t * torch.cos(half) / (2.0 * torch.sin(half))) / (t * t))
    I = torch.eye(3, dtype=phi.dtype).expand_as(Phi)
    return I - 0.5 * Phi + c2 * Phi2


# --------------------------------------------------------------------------- SE3 forward ops
def expm(a):
    """se3.h:134-142  SE3::Exp"""
    tau, phi = a[..., :3], a[..., 3:]
    q = so3_exp(phi)
    t = (so3_left_jacobian(phi) @ tau[..., None])[..., 0]
    return torch.cat([t, q], -1)


def logm(X):
    """se3.h:124-132  SE3::Log"""
    t, q = _split(X)
    phi = so3_log(q)
    tau = (so3_left_jacobian_inverse(phi) @ t[..., None])[..., 0]
    return torch.cat([tau, phi], -1)


def inv(X):
    """se3.h:36-38"""
    t, q = _split(X)
    qi = _qnormalize(_qconj(q))
    return torch.cat([-_qrot(qi, t), qi], -1)


def mul(X, Y):
    """se3.h:45-47"""
    tx, qx = _split(X)
    ty, qy = _split(Y)
    q = _qnormalize(_qmul(qx, qy))
    return torch.cat([tx + _qrot(qx, ty), q], -1)


def act(X, p):
    """se3.h:49-51"""
    t, q = _split(X)
    return _qrot(q, p) + t


def act4(X, p):
    """se3.h:53-56"""
    t, q = _split(X)
    return torch.cat([_qrot(q, p[..., :3]) + t * p[..., 3:], p[..., 3:]], -1)


def Adj(X):
    """se3.h:58-67  [[R, [t]x R], [0, R]]"""
    t, q = _split(X)
    R = _qmat(q)
    tR = _hat(t) @ R
    Z = torch.zeros_like(R)
    return torch.cat([torch.cat([R, tR], -1), torch.cat([Z, R], -1)], -2)


def adj(X, a):
    """se3.h:80-82"""
    return (Adj(X) @ a[..., None])[..., 0]


def adjT(X, a):
    """se3.h:84-86"""
    return (Adj(X).transpose(-1, -2) @ a[..., None])[..., 0]


def as_matrix(X):
    """se3.h:69-74, row-major 4x4 (lietorch_gpu.cu:258-269)"""
    t, q = _split(X)
    R = _qmat(q)
    top = torch.cat([R, t[..., None]], -1)
    bot = torch.zeros_like(top[..., :1, :])
    bot[..., 0, 3] = 1
    return torch.cat([top, bot], -2)


def small_adj(a):
    """se3.h:100-113  ad(tau,phi) = [[Phi, Tau],[0, Phi]]"""
    Tau, Phi = _hat(a[..., :3]), _hat(a[..., 3:])
    Z = torch.zeros_like(Phi)
    return torch.cat([torch.cat([Phi, Tau], -1), torch.cat([Z, Phi], -1)], -2)


def _calcQ(a):
    """se3.h:144-173"""
    tau, phi = a[..., :3], a[..., 3:]
    Tau, Phi = _hat(tau), _hat(phi)
    theta2 = (phi * phi).sum(-1)[..., None, None]
    theta = theta2.sqrt()
    theta4 = theta2 * theta2
    small = theta < EPS
    t = torch.where(small, torch.ones_like(theta), theta)
    t2, t4 = t * t, t * t * t * t
    c1 = torch.where(small, 1.0 / 6.0 - (1.0 / 120.0) * theta2, (t - torch.sin(t)) / (t2 * t))
    c2 = torch.where(small, 1.0 / 24.0 - (1.0 / 720.0) * theta2,
                     (t2 + 2 * torch.cos(t) - 2) / (2 * t4))
    c3 = torch.where(small, 1.0 / 120.0 - (1.0 / 2520.0) * theta2,
                     (2 * t - 3 * torch.sin(t) + t * torch.cos(t)) / (2 * t4 * t))
    PT, TP = Phi @ Tau, Tau @ Phi
    PTP = PT @ Phi
    return 0.5 * Tau + c1 * (PT + TP + PTP) \
        + c2 * (Phi @ PT + TP @ Phi - 3 * PTP) \
        + c3 * (PTP @ Phi + Phi @ PTP)


def left_jacobian(a):
    """se3.h:175-186"""
    J = so3_left_jacobian(a[..., 3:])
    Q = _calcQ(a)
    Z = torch.zeros_like(J)
    return torch.cat([torch.cat([J, Q], -1), torch.cat([Z, J], -1)], -2)


def left_jacobian_inverse(a):
    """se3.h:188-201"""
    Ji = so3_left_jacobian_inverse(a[..., 3:])
    Q = _calcQ(a)
    Z = torch.zeros_like(Ji)
    return torch.cat([torch.cat([Ji, -Ji @ Q @ Ji], -1), torch.cat([Z, Ji], -1)], -2)


def jinv(X, a):
    """lietorch_gpu.cu:283-294  J_l^{-1}(Log X) a"""
    return (left_jacobian_inverse(logm(X)) @ a[..., None])[..., 0]


# --------------------------------------------------------------------------- backward ops
def _pad7(g6):
    return torch.cat([g6, torch.zeros_like(g6[..., :1])], -1)


def _rowmat(g, M):
    return (g[..., None, :] @ M)[..., 0, :]


def expm_backward(grad, a):
    """lietorch_gpu.cu:32-44  da = dX * J_l(a)"""
    return _rowmat(grad[..., :6], left_jacobian(a))


def logm_backward(grad, X):
    """lietorch_gpu.cu:58-70  dX = da * J_l^{-1}(Log X)"""
    return _pad7(_rowmat(grad[..., :6], left_jacobian_inverse(logm(X))))


def inv_backward(grad, X):
    """lietorch_gpu.cu:85-97  dX = -dY * Adj(X^{-1})"""
    return _pad7(-_rowmat(grad[..., :6], Adj(inv(X))))


def mul_backward(grad, X, Y):
    """lietorch_gpu.cu:112-125  dX = dZ, dY = dZ * Adj(X)"""
    g = grad[..., :6]
    return _pad7(g.clone()), _pad7(_rowmat(g, Adj(X)))


def adj_backward(grad, X, a):
    """lietorch_gpu.cu:140-157"""
    A = Adj(X)
    b = (A @ a[..., None])[..., 0]
    return _pad7(-_rowmat(grad, small_adj(b))), _rowmat(grad, A)


def adjT_backward(grad, X, a):
    """lietorch_gpu.cu:173-188  da = Adj(X) db, dX = -a * ad(Adj(X) db)"""
    Adb = (Adj(X) @ grad[..., None])[..., 0]
    return _pad7(-_rowmat(a, small_adj(Adb))), Adb


def act_backward(grad, X, p):
    """lietorch_gpu.cu:204-221"""
    t, q = _split(X)
    R = _qmat(q)
    pp = act(X, p)
    I = torch.eye(3, dtype=X.dtype).expand(*pp.shape[:-1], 3, 3)
    J = torch.cat([I, _hat(-pp)], -1)
    return _pad7(_rowmat(grad, J)), _rowmat(grad, R)


def act4_backward(grad, X, p):
    """lietorch_gpu.cu:238-256, se3.h:211-217"""
    T = as_matrix(X)
    pp = act4(X, p)
    I = torch.eye(3, dtype=X.dtype).expand(*pp.shape[:-1], 3, 3)
    J = torch.cat([pp[..., 3:, None] * I, _hat(-pp[..., :3])], -1)
    J = torch.cat([J, torch.zeros_like(J[..., :1, :])], -2)
    return _pad7(_rowmat(grad, J)), _rowmat(grad, T)
