"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of the reference's Python geometry + differentiable bundle adjustment:
  iproj / proj / transform (+ Jacobians)   devo/projective_ops.py:19-105
  point_cloud, flow_mag                    devo/projective_ops.py:107-121
  BA (training / CPU-baseline variant)     devo/ba.py:86-182, CholeskySolver :12-37,
                                           scatter helpers :40-56, block_matmul/solve :58-77

Unlike the CUDA extensions these reference modules ARE importable in the build container
(with shims for torch_scatter / lietorch_backends): tools/gen_golden.py
runs the real reference functions on seeded inputs and tests/test_oracle_golden.py pins
this restatement against those outputs (tests/golden/*.npz).

The restatement is written in closed form (Jacobian rows expanded by hand, the normal
equations assembled as dense 2-D matrices) rather than as the reference's chain of 5-D block
tensors; the arithmetic is the same up to fp rounding.  `torch_scatter.scatter_sum` (v2.0.9,
environment.yml:107 — third-party, absent from /root/reference) is a plain segmented sum and
is restated with index_add / index_put(accumulate).
"""
import torch
from .lie import SE3  # noqa: F401  (re-exported for tests)

MIN_DEPTH = 0.2


def _K(intrinsics):
    """fx, fy, cx, cy broadcast over the PxP patch grid (projective_ops.py:22,36)."""
    return [intrinsics[..., k, None, None] for k in range(4)]


def iproj(patches, intrinsics):
    """projective_ops.py:19-29: pixel (x, y, inverse depth d) -> homogeneous (X, Y, 1, d)."""
    fx, fy, cx, cy = _K(intrinsics)
    px, py, pd = patches[:, :, 0], patches[:, :, 1], patches[:, :, 2]
    return torch.stack([(px - cx) / fx, (py - cy) / fy, torch.ones_like(pd), pd], dim=-1)


def proj(Xh, intrinsics, depth=False):
    """projective_ops.py:32-50: pinhole projection with Z clamped at 0.1 (:43)."""
    fx, fy, cx, cy = _K(intrinsics)
    rz = 1.0 / Xh[..., 2].clamp(min=0.1)
    u = fx * (rz * Xh[..., 0]) + cx
    v = fy * (rz * Xh[..., 1]) + cy
    return torch.stack([u, v, rz] if depth else [u, v], dim=-1)


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False, tonly=False):
    """projective_ops.py:53-105.  poses: oracle SE3 [1,n,7]; patches [1,Np,3,P,P].
    Returns coords [1,E,P,P,2] (+ validity, + (Ji [1,E,2,6], Jj [1,E,2,6], Jz [1,E,2,1]))."""
    Gij = poses[:, jj] * poses[:, ii].inv()                        # :61
    if tonly:                                                      # :63-64
        Gij.data[..., 3:] = torch.as_tensor([0, 0, 0, 1], dtype=Gij.data.dtype)
    X1 = Gij[:, :, None, None] * iproj(patches[:, kk], intrinsics[:, ii])     # :57,66
    c = X1.shape[2] // 2
    x1 = proj(X1, intrinsics[:, jj], depth)                        # :70

    if jacobian:                                                   # :73-98
        X, Y, Z, H = X1[..., c, c, :].unbind(dim=-1)
        fx, fy = intrinsics[:, jj, 0], intrinsics[:, jj, 1]
        o = torch.zeros_like(Z)
        d = torch.where(Z.abs() > 0.2, 1.0 / torch.where(Z.abs() > 0.2, Z, torch.ones_like(Z)), o)   # :80-81
        # Jj = Jp @ Ja (:83-95) expanded row by row
        Jj = torch.stack([
            torch.stack([fx * d * H, o, -fx * X * d * d * H, -fx * X * d * d * Y,
                         fx * d * Z + fx * X * d * d * X, -fx * d * Y], -1),
            torch.stack([o, fy * d * H, -fy * Y * d * d * H, -fy * d * Z - fy * Y * d * d * Y,
                         fy * Y * d * d * X, fy * d * X], -1)], dim=-2)
        Ji = -Gij[:, :, None].adjT(Jj)                             # :96
        t = Gij.matrix()[..., :3, 3]                               # :97 reads the 4th column of Gij.matrix(): the same VALUE as Gij.data[..., :3], but the
        #                                                            gradient reaches Gij through the group action (tangent convention, rotation part included)
        Jz = torch.stack([fx * d * t[..., 0] - fx * X * d * d * t[..., 2],
                          fy * d * t[..., 1] - fy * Y * d * d * t[..., 2]], -1)[..., None]
        return x1, (Z > 0.2).to(Z.dtype), (Ji, Jj, Jz)             # :100

    if valid:                                                      # :102-103
        return x1, (X1[..., c, c, 2] > 0.2).to(x1.dtype)
    return x1


def point_cloud(poses, patches, intrinsics, ix):
    """projective_ops.py:107-109"""
    return poses[:, ix, None, None].inv() * iproj(patches, intrinsics[:, ix])


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3):
    """projective_ops.py:111-121"""
    c0 = transform(poses, patches, intrinsics, ii, ii, kk)
    full = (transform(poses, patches, intrinsics, ii, jj, kk) - c0).norm(dim=-1)
    trans = (transform(poses, patches, intrinsics, ii, jj, kk, tonly=True) - c0).norm(dim=-1)
    return beta * full + (1 - beta) * trans


# --------------------------------------------------------------------------- devo/ba.py
class CholeskySolver(torch.autograd.Function):
    """ba.py:12-37: x = H^{-1} b by Cholesky; zeros (and no gradient) if the factorisation fails;
    backward: dz = H^{-1} g, dH = -x dz^T."""
    @staticmethod
    def forward(ctx, H, b):
        L, info = torch.linalg.cholesky_ex(H)
        ctx.failed = bool(torch.any(info))
        if ctx.failed:
            return torch.zeros_like(b)
        x = torch.cholesky_solve(b, L)
        ctx.save_for_backward(L, x)
        return x

    @staticmethod
    def backward(ctx, g):
        if ctx.failed:
            return None, None
        L, x = ctx.saved_tensors
        dz = torch.cholesky_solve(g, L)
        return -x @ dz.transpose(-1, -2), dz


def _acc2d(shape, rows, cols, vals, dtype):
    """dense accumulate: out[rows, cols] += vals (the scatter_sum of ba.py:40-47)."""
    return torch.zeros(shape, dtype=dtype).index_put((rows, cols), vals, accumulate=True)


def BA(poses, patches, intrinsics, targets, weights, lmbda, ii, jj, kk, bounds, ep=100.0,
       fixedp=1, structure_only=False):
    """ba.py:86-182 (ONE Gauss-Newton step, differentiable).  Shapes: batch 1 only.
    Returns (poses: oracle SE3, patches)."""
    dt = patches.dtype
    n = max(int(ii.max()), int(jj.max())) + 1 - fixedp             # :90,133
    coords, ok, (Ji, Jj, Jz) = transform(poses, patches, intrinsics, ii, jj, kk, jacobian=True)   # :92-93
    c = coords.shape[3] // 2
    ctr = coords[0, :, c, c, :]                                    # [E,2]
    r = targets[0] - ctr                                           # :96
    gate = ok[0] * (r.norm(dim=-1) < 250).to(dt)                   # :98
    gate = gate * ((ctr[:, 0] > bounds[0]) & (ctr[:, 1] > bounds[1]) &
                   (ctr[:, 0] < bounds[2]) & (ctr[:, 1] < bounds[3])).to(dt)       # :100-106
    r = gate[:, None] * r                                          # :111
    w = gate[:, None] * weights[0]                                 # :112
    Ji, Jj, Jz = Ji[0], Jj[0], Jz[0, :, :, 0]                      # [E,2,6], [E,2,6], [E,2]

    kx, ku = torch.unique(kk, return_inverse=True, sorted=True)    # :137
    m = kx.shape[0]
    a, b_ = ii - fixedp, jj - fixedp                               # :134-135
    six = torch.arange(6)

    # per-edge 6x6 / 6x1 blocks, sum over the two residual rows (:114-127)
    def outer(A, Bm):
        return torch.einsum('er,erp,erq->epq', w, A, Bm)

    dense = torch.zeros(6 * max(n, 0), 6 * max(n, 0), dtype=dt)
    Emat = torch.zeros(6 * max(n, 0), m, dtype=dt)
    vvec = torch.zeros(6 * max(n, 0), dtype=dt)
    if n > 0:
        for (ra, Ja), (rb, Jb) in (((a, Ji), (a, Ji)), ((a, Ji), (b_, Jj)), ((b_, Jj), (a, Ji)), ((b_, Jj), (b_, Jj))):
            sel = (ra >= 0) & (rb >= 0) & (ra < n) & (rb < n)      # :40-42
            rows = (6 * ra[sel])[:, None, None] + six[None, :, None]
            cols = (6 * rb[sel])[:, None, None] + six[None, None, :]
            blk = outer(Ja, Jb)[sel]
            dense = dense + _acc2d(dense.shape, rows.expand_as(blk), cols.expand_as(blk), blk, dt)    # :139-142
        for ra, Ja in ((a, Ji), (b_, Jj)):
            sel = (ra >= 0) & (ra < n)
            rows = (6 * ra[sel])[:, None] + six[None]
            eblk = torch.einsum('er,erp,er->ep', w, Ja, Jz)[sel]
            Emat = Emat + _acc2d(Emat.shape, rows, ku[sel][:, None].expand_as(rows), eblk, dt)        # :144-145
            vblk = torch.einsum('er,erp,er->ep', w, Ja, r)[sel]
            vvec = vvec + torch.zeros_like(vvec).index_put((rows,), vblk, accumulate=True)            # :149-150
    C = torch.zeros(m, dtype=dt).index_add(0, ku, (w * Jz * Jz).sum(-1))                              # :147
    u = torch.zeros(m, dtype=dt).index_add(0, ku, (w * Jz * r).sum(-1))                               # :152

    if isinstance(lmbda, torch.Tensor):
        lmbda = lmbda.reshape(m)                                   # :155-156 per-patch damping
    Q = 1.0 / (C + lmbda)                                          # :157

    if structure_only or n == 0:                                   # :162-163
        dZ = Q * u
        dX = None
    else:
        EQ = Emat * Q[None]                                        # :160
        S = dense - EQ @ Emat.t()                                  # :166
        y = vvec - EQ @ u                                          # :167
        A = S + (ep + 1e-4 * S) * torch.eye(6 * n, dtype=dt)       # :73  (diagonal damping)
        dX = CholeskySolver.apply(A[None], y[None, :, None])[0, :, 0]          # :75
        dZ = Q * (u - Emat.t() @ dX)                               # :170

    disp = patches[:, :, 2] + torch.zeros_like(patches[:, :, 2]).index_add(
        1, kx, dZ.view(1, -1, 1, 1).expand(1, m, *patches.shape[-2:]))
    disp = disp.clamp(min=1e-3, max=10.0)                          # :175-176
    patches = torch.stack([patches[:, :, 0], patches[:, :, 1], disp], dim=2)
    if dX is not None:                                             # :179-180
        upd = torch.zeros(1, poses.data.shape[1], 6, dtype=dt)
        upd = upd.index_add(1, fixedp + torch.arange(n), dX.view(1, n, 6))
        poses = poses.retr(upd)
    return poses, patches
