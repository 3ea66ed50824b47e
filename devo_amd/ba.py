"""Differentiable bundle adjustment for training — host-side mirror of devo/ba.py:86-182 (one
Gauss-Newton step per call, constants of SURVEY.md Appendix B: residual gate 250 px, explicit bounds,
damping ep + 1e-4*diag, depth clamp [1e-3, 10], first `fixedp` poses fixed).

The reference builds the normal equations with ten batched matmuls and ten torch_scatter.scatter_sum calls
into 5-D block tensors; torch_scatter has no ROCm wheel here, so the system is assembled directly as dense
2-D matrices with index_put_(accumulate=True) — same arithmetic, autograd-complete.  Group maths and the
reprojection run on the HIP SE3 kernels through devo_amd.lietorch.
"""
import torch
from . import projective_ops as pops


class CholeskySolver(torch.autograd.Function):
    """devo/ba.py:12-37: solve by Cholesky; zeros and no gradient when the factorisation fails."""
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


def BA(poses, patches, intrinsics, targets, weights, lmbda, ii, jj, kk, bounds, ep=100.0, PRINT=False,
       fixedp=1, structure_only=False):
    dev, dt = patches.device, patches.dtype
    n = max(int(ii.max()), int(jj.max())) + 1 - fixedp
    coords, ok, (Ji, Jj, Jz) = pops.transform(poses, patches, intrinsics, ii, jj, kk, jacobian=True)
    c = coords.shape[3] // 2
    ctr = coords[0, :, c, c, :]
    r = targets[0] - ctr
    gate = ok[0] * (r.norm(dim=-1) < 250).to(dt)
    gate = gate * ((ctr[:, 0] > bounds[0]) & (ctr[:, 1] > bounds[1]) &
                   (ctr[:, 0] < bounds[2]) & (ctr[:, 1] < bounds[3])).to(dt)
    if PRINT:
        print((r * gate[:, None]).norm(dim=-1).mean().item())
    r = gate[:, None] * r
    w = gate[:, None] * weights[0]
    Ji, Jj, Jz = Ji[0], Jj[0], Jz[0, :, :, 0]

    kx, ku = torch.unique(kk, return_inverse=True, sorted=True)
    m = kx.shape[0]
    a, b_ = ii - fixedp, jj - fixedp
    six = torch.arange(6, device=dev)
    n6 = 6 * max(n, 0)
    S = torch.zeros(n6, n6, dtype=dt, device=dev)
    Emat = torch.zeros(n6, m, dtype=dt, device=dev)
    v = torch.zeros(n6, dtype=dt, device=dev)
    if n > 0:
        for (ra, Ja), (rb, Jb) in (((a, Ji), (a, Ji)), ((a, Ji), (b_, Jj)), ((b_, Jj), (a, Ji)), ((b_, Jj), (b_, Jj))):
            sel = (ra >= 0) & (rb >= 0) & (ra < n) & (rb < n)
            blk = torch.einsum('er,erp,erq->epq', w, Ja, Jb)[sel]
            rows = ((6 * ra[sel])[:, None, None] + six[None, :, None]).expand_as(blk)
            cols = ((6 * rb[sel])[:, None, None] + six[None, None, :]).expand_as(blk)
            S = S.index_put((rows, cols), blk, accumulate=True)
        for ra, Ja in ((a, Ji), (b_, Jj)):
            sel = (ra >= 0) & (ra < n)
            rows = (6 * ra[sel])[:, None] + six[None]
            Emat = Emat.index_put((rows, ku[sel][:, None].expand_as(rows)),
                                  torch.einsum('er,erp,er->ep', w, Ja, Jz)[sel], accumulate=True)
            v = v.index_put((rows,), torch.einsum('er,erp,er->ep', w, Ja, r)[sel], accumulate=True)
    C = torch.zeros(m, dtype=dt, device=dev).index_add(0, ku, (w * Jz * Jz).sum(-1))
    u = torch.zeros(m, dtype=dt, device=dev).index_add(0, ku, (w * Jz * r).sum(-1))

    if isinstance(lmbda, torch.Tensor):
        lmbda = lmbda.reshape(m)
    Q = 1.0 / (C + lmbda)

    if structure_only or n == 0:
        dZ, dX = Q * u, None
    else:
        EQ = Emat * Q[None]
        A = S - EQ @ Emat.t()
        y = v - EQ @ u
        A = A + (ep + 1e-4 * A) * torch.eye(n6, dtype=dt, device=dev)
        dX = CholeskySolver.apply(A[None], y[None, :, None])[0, :, 0]
        dZ = Q * (u - Emat.t() @ dX)

    P = patches.shape[-1]
    disp = patches[:, :, 2] + torch.zeros_like(patches[:, :, 2]).index_add(1, kx, dZ.view(1, m, 1, 1).expand(1, m, P, P))
    patches = torch.stack([patches[:, :, 0], patches[:, :, 1], disp.clamp(min=1e-3, max=10.0)], dim=2)
    if dX is not None:
        upd = torch.zeros(1, poses.data.shape[1], 6, dtype=dt, device=dev)
        upd = upd.index_add(1, fixedp + torch.arange(n, device=dev), dX.view(1, n, 6))
        poses = poses.retr(upd)
    return poses, patches
