"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of the reference's `cuda_ba` extension (fastba), i.e. the NON-differentiable
inference bundle adjustment and its helpers:
  SE3 device maths      devo/fastba/ba_cuda.cu:18-156 (actSO3, actSE3, adjSE3, relSE3, expSO3, expSE3, retrSE3)
  residuals + Hessian   ba_cuda.cu:214-365
  Schur / solve / loop  ba_cuda.cu:422-540
  pose / patch retract  ba_cuda.cu:160-211
  reproject             ba_cuda.cu:368-418, 543-575
  neighbors             devo/fastba/ba.cpp:104-149

The CUDA source cannot run here (no GPU) and ba.cpp needs libtorch + the CUDA half of the
extension to link, so it is unbuildable with gcc alone: "parity unpinned" by reference
output.  Pinned by hand-solvable known-answer tests (tests/test_oracle_fastba.py), by an
fp64 run of the same restatement, and cross-checked against the reference's *Python* BA
(devo/ba.py, imported with shims by tools/gen_golden.py) on problems where the two
variants coincide (SURVEY.md Appendix B lists the differences).

All maths is written edge-vectorised in torch on CPU; `dtype` selects fp32 (mirrors the
CUDA arithmetic) or fp64 (ground truth).
"""
import torch


# --------------------------------------------------------------------------- SE3 device maths
def act_so3(q, X):
    """ba_cuda.cu:18-28"""
    qv, w = q[..., :3], q[..., 3:]
    uv = 2.0 * torch.linalg.cross(qv, X, dim=-1)
    return X + w * uv + torch.linalg.cross(qv, uv, dim=-1)


def act_se3(t, q, X):
    """ba_cuda.cu:30-37  (X is homogeneous [x,y,z,w])"""
    Y = act_so3(q, X[..., :3]) + X[..., 3:] * t
    return torch.cat([Y, X[..., 3:]], -1)


def adj_se3(t, q, X):
    """ba_cuda.cu:39-54  Y = Adj(t,q)^T X for a 6-vector X"""
    qinv = torch.cat([-q[..., :3], q[..., 3:]], -1)
    Y0 = act_so3(qinv, X[..., :3])
    Y1 = act_so3(qinv, X[..., 3:])
    a, b = X[..., :3], t
    u = torch.stack([b[..., 2] * a[..., 1] - b[..., 1] * a[..., 2],
                     b[..., 0] * a[..., 2] - b[..., 2] * a[..., 0],
                     b[..., 1] * a[..., 0] - b[..., 0] * a[..., 1]], -1)
    return torch.cat([Y0, Y1 + act_so3(qinv, u)], -1)


def rel_se3(ti, qi, tj, qj):
    """ba_cuda.cu:56-67  G_ij = G_j * G_i^{-1}"""
    x = -qj[..., 3] * qi[..., 0] + qj[..., 0] * qi[..., 3] - qj[..., 1] * qi[..., 2] + qj[..., 2] * qi[..., 1]
    y = -qj[..., 3] * qi[..., 1] + qj[..., 1] * qi[..., 3] - qj[..., 2] * qi[..., 0] + qj[..., 0] * qi[..., 2]
    z = -qj[..., 3] * qi[..., 2] + qj[..., 2] * qi[..., 3] - qj[..., 0] * qi[..., 1] + qj[..., 1] * qi[..., 0]
    w = qj[..., 3] * qi[..., 3] + qj[..., 0] * qi[..., 0] + qj[..., 1] * qi[..., 1] + qj[..., 2] * qi[..., 2]
    qij = torch.stack([x, y, z, w], -1)
    tij = tj - act_so3(qij, ti)
    return tij, qij


def exp_so3(phi):
    """ba_cuda.cu:70-92"""
    theta_sq = (phi * phi).sum(-1, keepdim=True)
    theta_p4 = theta_sq * theta_sq
    theta = theta_sq.sqrt()
    small = theta_sq < 1e-8
    ts = torch.where(small, torch.ones_like(theta), theta)
    imag = torch.where(small, 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_p4, torch.sin(0.5 * ts) / ts)
    real = torch.where(small, 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * theta_p4, torch.cos(0.5 * ts))
    return torch.cat([imag * phi, real], -1)


def exp_se3(xi):
    """ba_cuda.cu:108-135"""
    tau, phi = xi[..., :3], xi[..., 3:]
    q = exp_so3(phi)
    theta_sq = (phi * phi).sum(-1, keepdim=True)
    theta = theta_sq.sqrt()
    big = theta > 1e-4
    ts = torch.where(big, theta, torch.ones_like(theta))
    tsq = torch.where(big, theta_sq, torch.ones_like(theta))
    a = (1 - torch.cos(ts)) / tsq
    b = (ts - torch.sin(ts)) / (ts * tsq)
    c1 = torch.linalg.cross(phi, tau, dim=-1)
    c2 = torch.linalg.cross(phi, c1, dim=-1)
    t = tau + torch.where(big, a * c1 + b * c2, torch.zeros_like(tau))
    return t, q


def retr_se3(xi, t, q):
    """ba_cuda.cu:138-156  (dt,dq)=Exp(xi); q1 = dq*q; t1 = dq*t + dt   (no renormalisation)"""
    dt, dq = exp_se3(xi)
    q1 = torch.stack([
        dq[..., 3] * q[..., 0] + dq[..., 0] * q[..., 3] + dq[..., 1] * q[..., 2] - dq[..., 2] * q[..., 1],
        dq[..., 3] * q[..., 1] + dq[..., 1] * q[..., 3] + dq[..., 2] * q[..., 0] - dq[..., 0] * q[..., 2],
        dq[..., 3] * q[..., 2] + dq[..., 2] * q[..., 3] + dq[..., 0] * q[..., 1] - dq[..., 1] * q[..., 0],
        dq[..., 3] * q[..., 3] - dq[..., 0] * q[..., 0] - dq[..., 1] * q[..., 1] - dq[..., 2] * q[..., 2]], -1)
    t1 = act_so3(dq, t) + dt
    return t1, q1


# --------------------------------------------------------------------------- kernels
def residuals_and_jacobians(poses, patches, intr, target, weight, ii, jj, kk):
    """Per-edge part of ba_cuda.cu:239-330: returns dict of r[E,2], w[E,2] (mask folded in),
    Ji[E,2,6], Jj[E,2,6], Jz[E,2].  poses [Nbuf,7], patches [Np,3,P,P], intr = row 0."""
    fx, fy, cx, cy = [intr[i] for i in range(4)]
    ti, qi = poses[ii, :3], poses[ii, 3:]
    tj, qj = poses[jj, :3], poses[jj, 3:]
    px, py, pd = patches[kk, 0, 1, 1], patches[kk, 1, 1, 1], patches[kk, 2, 1, 1]
    Xi = torch.stack([(px - cx) / fx, (py - cy) / fy, torch.ones_like(px), pd], -1)
    tij, qij = rel_se3(ti, qi, tj, qj)
    Xj = act_se3(tij, qij, Xi)
    X, Y, Z, W = Xj.unbind(-1)
    d = torch.where(Z >= 0.2, 1.0 / Z, torch.zeros_like(Z))
    d2 = d * d
    x1 = fx * (X / Z) + cx
    y1 = fy * (Y / Z) + cy
    rx = target[:, 0] - x1
    ry = target[:, 1] - y1
    inb = (torch.sqrt(rx * rx + ry * ry) < 128) & (Z > 0.2) & (x1 > -64) & (y1 > -64) \
        & (x1 < 2 * cx + 64) & (y1 < 2 * cy + 64)
    mask = inb.to(poses.dtype)
    o = torch.zeros_like(X)
    Jjx = torch.stack([fx * W * d, o, fx * -X * W * d2, fx * -X * Y * d2, fx * (1 + X * X * d2), fx * -Y * d], -1)
    Jjy = torch.stack([o, fy * W * d, fy * -Y * W * d2, fy * (-1 - Y * Y * d2), fy * (X * Y * d2), fy * X * d], -1)
    Jzx = fx * (tij[:, 0] * d - tij[:, 2] * (X * d2))
    Jzy = fy * (tij[:, 1] * d - tij[:, 2] * (Y * d2))
    Jix = adj_se3(tij, qij, Jjx)
    Jiy = adj_se3(tij, qij, Jjy)
    return dict(r=torch.stack([rx, ry], -1), w=mask[:, None] * weight,
                Ji=torch.stack([Jix, Jiy], 1), Jj=torch.stack([Jjx, Jjy], 1),
                Jz=torch.stack([Jzx, Jzy], -1), mask=mask, x1=x1, y1=y1, Z=Z)


def unique_kk(kk):
    """torch::_unique(kk, sorted, return_inverse)  (ba_cuda.cu:435-437) -> kx, ku  (integer, bit-exact)"""
    kx, ku = torch.unique(kk, sorted=True, return_inverse=True)
    return kx, ku


def build_system(J, ii, jj, ku, t0, N, M, dtype):
    """Accumulation part of ba_cuda.cu:294-362 (dense B,E,C,v,u)."""
    B = torch.zeros(6 * N, 6 * N, dtype=dtype)
    E = torch.zeros(6 * N, M, dtype=dtype)
    C = torch.zeros(M, dtype=dtype)
    v = torch.zeros(6 * N, dtype=dtype)
    u = torch.zeros(M, dtype=dtype)
    ix, jx = ii - t0, jj - t0
    ar = torch.arange(6)
    for row in range(2):
        w, r = J["w"][:, row], J["r"][:, row]
        Ji, Jj, Jz = J["Ji"][:, row], J["Jj"][:, row], J["Jz"][:, row]

        def add_block(a, bidx, Ja, Jb, sign, sel):
            blk = sign * w[sel, None, None] * Ja[sel, :, None] * Jb[sel, None, :]      # [e,6,6]
            rows = (6 * a[sel])[:, None, None] + ar[None, :, None]
            cols = (6 * bidx[sel])[:, None, None] + ar[None, None, :]
            B.index_put_((rows.expand_as(blk), cols.expand_as(blk)), blk, accumulate=True)

        si, sj = ix >= 0, jx >= 0
        add_block(ix, ix, Ji, Ji, 1.0, si)
        add_block(jx, jx, Jj, Jj, 1.0, sj)
        add_block(ix, jx, Ji, Jj, -1.0, si & sj)
        add_block(jx, ix, Jj, Ji, -1.0, si & sj)
        rows_i = (6 * ix[si])[:, None] + ar[None]
        rows_j = (6 * jx[sj])[:, None] + ar[None]
        E.index_put_((rows_i, ku[si][:, None].expand_as(rows_i)), -(w * Jz)[si, None] * Ji[si], accumulate=True)
        E.index_put_((rows_j, ku[sj][:, None].expand_as(rows_j)), (w * Jz)[sj, None] * Jj[sj], accumulate=True)
        v.index_put_((rows_i,), -(w * r)[si, None] * Ji[si], accumulate=True)
        v.index_put_((rows_j,), (w * r)[sj, None] * Jj[sj], accumulate=True)
        C.index_add_(0, ku, w * Jz * Jz)
        u.index_add_(0, ku, w * r * Jz)
    return B, E, C, v, u


def ba(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2,
       dtype=torch.float32, return_system=False):
    """cuda_ba.forward (ba_cuda.cu:422-540).  Returns NEW (poses, patches) tensors with the
    shapes of the inputs (the reference mutates in place; the oracle is functional)."""
    pshape, qshape = poses.shape, patches.shape
    P = patches.shape[-1]
    poses = poses.reshape(-1, 7).to(dtype).clone()
    patches = patches.reshape(-1, 3, P, P).to(dtype).clone()
    intr = intrinsics.reshape(-1, 4)[0].to(dtype)            # only row 0 is read (ba_cuda.cu:233-237)
    target = target.reshape(-1, 2).to(dtype)
    weight = weight.reshape(-1, 2).to(dtype)
    lm = lmbda.reshape(-1)[0].to(dtype)
    kx, ku = unique_kk(kk)
    N, M = t1 - t0, kx.shape[0]
    sys_ = None
    for _ in range(iterations):
        J = residuals_and_jacobians(poses, patches, intr, target, weight, ii, jj, kk)
        B, E, C, v, u = build_system(J, ii, jj, ku, t0, N, M, dtype)
        Q = 1.0 / (C + lm)
        if N == 0:
            dZ = Q * u
            dX = None
        else:
            EQ = E * Q[None]
            S = B - EQ @ E.t()
            y = v - EQ @ u
            S = S + torch.eye(6 * N, dtype=dtype) * (1e-4 * S + 1.0)
            U = torch.linalg.cholesky(S)
            dX = torch.cholesky_solve(y[:, None], U)[:, 0]
            dZ = Q * (u - E.t() @ dX)
            if sys_ is None:
                sys_ = dict(B=B, E=E, C=C, v=v, u=u, S=S, y=y, dX=dX.clone(), dZ=dZ.clone())
            tn, qn = retr_se3(dX.view(N, 6), poses[t0:t1, :3], poses[t0:t1, 3:])
            poses[t0:t1] = torch.cat([tn, qn], -1)
        # patch_retr_kernel (ba_cuda.cu:191-211): reads pixel [0][0], writes all PxP
        d = patches[kx, 2, 0, 0] + dZ
        d = torch.where(d > 20, torch.ones_like(d), d)
        d = torch.clamp(d, min=1e-4)
        patches[kx, 2] = d[:, None, None].expand(-1, P, P)
    out = (poses.view(pshape), patches.view(qshape))
    return out + (sys_,) if return_system else out


def reproject(poses, patches, intrinsics, ii, jj, kk, dtype=torch.float32):
    """cuda_ba.reproject (ba_cuda.cu:368-418, 543-575) -> coords [1,E,2,P,P]; no Z clamp."""
    P = patches.shape[-1]
    poses = poses.reshape(-1, 7).to(dtype)
    patches = patches.reshape(-1, 3, P, P).to(dtype)
    fx, fy, cx, cy = [intrinsics.reshape(-1, 4)[0, i].to(dtype) for i in range(4)]
    tij, qij = rel_se3(poses[ii, :3], poses[ii, 3:], poses[jj, :3], poses[jj, 3:])
    pk = patches[kk]                                               # [E,3,P,P]
    Xi = torch.stack([(pk[:, 0] - cx) / fx, (pk[:, 1] - cy) / fy, torch.ones_like(pk[:, 0]), pk[:, 2]], -1)
    Xj = act_se3(tij[:, None, None], qij[:, None, None], Xi)
    x = fx * (Xj[..., 0] / Xj[..., 2]) + cx
    y = fy * (Xj[..., 1] / Xj[..., 2]) + cy
    return torch.stack([x, y], 1)[None]


def neighbors(ii, jj):
    """cuda_ba.neighbors (ba.cpp:104-149): per unique ii, stable-sort its edges by jj; ix = previous
    edge, jx = next edge, -1 at the ends.  Integer, bit-exact."""
    E = ii.shape[0]
    uniq, perm = torch.unique(ii, sorted=True, return_inverse=True)
    index = [[] for _ in range(uniq.shape[0])]
    pl = perm.tolist()
    for e in range(E):
        index[pl[e]].append(e)
    jl = jj.tolist()
    ix = torch.empty(E, dtype=torch.int64)
    jx = torch.empty(E, dtype=torch.int64)
    for idx in index:
        idx = sorted(idx, key=lambda e: jl[e])                     # python sort is stable
        for n, e in enumerate(idx):
            ix[e] = idx[n - 1] if n > 0 else -1
            jx[e] = idx[n + 1] if n < len(idx) - 1 else -1
    return ix, jx
