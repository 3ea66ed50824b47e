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
    """devo/ba.py:12-37: solve by Cholesky; zeros and no gradient when the factorisation fails.  The failure flag stays
    on the device (no host synchronisation): the result and the gradients are masked with it."""
    @staticmethod
    def forward(ctx, H, b):
        L, info = torch.linalg.cholesky_ex(H)
        bad = (info != 0).view(-1, 1, 1)
        x = torch.cholesky_solve(b, torch.where(bad, torch.eye(H.shape[-1], dtype=H.dtype, device=H.device), L))
        x = torch.where(bad, torch.zeros_like(x), x)
        ctx.save_for_backward(L, x, bad)
        return x

    @staticmethod
    def backward(ctx, g):
        L, x, bad = ctx.saved_tensors
        Ls = torch.where(bad, torch.eye(L.shape[-1], dtype=L.dtype, device=L.device), L)
        dz = torch.where(bad, torch.zeros_like(g), torch.cholesky_solve(g, Ls))
        return -x @ dz.transpose(-1, -2), dz


class _FusedSolve(torch.autograd.Function):
    """(edge terms [E,30]) -> (dX [6n], dZ [Np]): normal equations, Schur complement and damped Cholesky solve of ba.py:108-170 as
    ONE group of HIP kernels (devo_ba_solve_terms: the inference BA's accumulate / reduce / solve kernels fed with the given
    terms), and their adjoint (devo_ba_solve_terms_backward: one more solve with the saved matrix + per-patch / per-edge
    kernels) instead of ~35 + ~70 ATen kernels.  The Jacobians themselves and the retraction stay in the autograd graph."""

    @staticmethod
    def forward(ctx, terms, lm, ii, jj, kk, n_slots, t0, n_opt, ep):
        from .backends import cuda_ba
        dX, dZ, ws = cuda_ba.solve_terms(terms, lm, ii, jj, kk, n_slots, t0, n_opt, ep)
        ctx.save_for_backward(terms, ii, jj, kk)
        ctx.ws, ctx.meta = ws, (int(n_slots), int(t0), int(n_opt))
        return dX, dZ

    @staticmethod
    def backward(ctx, g_dX, g_dZ):
        from .backends import cuda_ba
        terms, ii, jj, kk = ctx.saved_tensors
        n_slots, t0, n_opt = ctx.meta
        g = cuda_ba.solve_terms_backward(terms, ii, jj, kk, n_slots, t0, n_opt, ctx.ws, g_dX, g_dZ)
        return g, None, None, None, None, None, None, None, None


class _EdgeTerms(torch.autograd.Function):
    """devo/ba.py:95-106 (residual, gates, the Jacobian blocks packed for the solve) in one HIP kernel per direction instead of ~15
    elementwise kernels and their adjoints (devo_ba_edge_terms / _backward).  The gate is piecewise constant: no gradient through it."""

    @staticmethod
    def forward(ctx, coords, valid, Ji, Jj, Jz, targets, weights, bounds):
        from .backends import cuda_ba
        terms, gate = cuda_ba.edge_terms(coords, valid, Ji, Jj, Jz, targets, weights, bounds)
        ctx.save_for_backward(gate)
        ctx.P = coords.shape[2]
        return terms

    @staticmethod
    def backward(ctx, g):
        from .backends import cuda_ba
        gate, = ctx.saved_tensors
        gc, gt, gw, gi, gj, gz = cuda_ba.edge_terms_backward(g, gate, ctx.P)
        return gc, None, gi, gj, gz, gt, gw, None


def _fused_path(patches, lmbda, n):
    import os
    per_patch = isinstance(lmbda, torch.Tensor) and lmbda.numel() > 1
    return (patches.is_cuda and patches.dtype == torch.float32 and not per_patch and n <= 32 and patches.shape[0] == 1
            and os.environ.get("DEVO_BA_TORCH", "0") != "1")


_lm_cache = {}


def _fused_step(poses, patches, terms, lmbda, ii, jj, kk, fixedp, n_opt, ep):
    """solve (HIP, differentiable) + depth update / clamp + pose retraction (devo/ba.py:159-182)"""
    dev, dt = patches.device, patches.dtype
    if isinstance(lmbda, torch.Tensor):
        lm = lmbda.reshape(1).to(dev, torch.float32)
    else:                                                              # a python number: one device scalar per (device, value), not a fill per step
        lm = _lm_cache.get((dev, float(lmbda)))
        if lm is None:
            lm = _lm_cache[(dev, float(lmbda))] = torch.full((1,), float(lmbda), dtype=torch.float32, device=dev)
    dX, dZ = _FusedSolve.apply(terms, lm, ii, jj, kk, patches.shape[1], fixedp, n_opt, float(ep))
    if poses.data.dtype == torch.float32 and poses.data.shape[0] == 1:
        new_poses, patches = _ApplyStep.apply(poses.data, patches, dX, dZ, int(fixedp), int(n_opt))      # retraction + depth update: one kernel
        return type(poses)(new_poses), patches
    disp = patches[:, :, 2] + dZ.view(1, -1, 1, 1)
    patches = torch.stack([patches[:, :, 0], patches[:, :, 1], disp.clamp(min=1e-3, max=10.0)], dim=2)
    if n_opt > 0:
        upd = torch.zeros(1, poses.data.shape[1], 6, dtype=dt, device=dev)
        upd[:, fixedp:fixedp + n_opt] = dX.view(1, n_opt, 6)
        poses = poses.retr(upd)
    return poses, patches


class _ApplyStep(torch.autograd.Function):
    """devo/ba.py:172-182 — depth update + clamp [1e-3, 10] + torch.stack, `poses.retr` on the optimised window — as one HIP kernel per
    direction (devo_ba_apply_step / _backward) instead of ~8 + ~12 ATen / SE3 launches."""

    @staticmethod
    def forward(ctx, poses, patches, dX, dZ, fixedp, n_opt):
        from . import _lib as L
        p7, q = poses.contiguous(), patches.contiguous()
        dXc, dZc = dX.contiguous(), dZ.contiguous()
        N, Np, P = p7.shape[1], q.shape[1], q.shape[-1]
        po, qo = torch.empty_like(p7), torch.empty_like(q)
        L.check(L.lib().devo_ba_apply_step(L.ptr(p7), L.ptr(q), L.ptr(dXc) if n_opt > 0 else None, L.ptr(dZc), N, Np, P, fixedp, n_opt, 1e-3, 10.0,
                                           L.ptr(po), L.ptr(qo), L.stream()), "ba.apply_step")
        ctx.save_for_backward(p7, q, dXc, dZc)
        ctx.meta = (N, Np, P, fixedp, n_opt)
        return po, qo

    @staticmethod
    def backward(ctx, g_poses, g_patches):
        from . import _lib as L
        p7, q, dXc, dZc = ctx.saved_tensors
        N, Np, P, fixedp, n_opt = ctx.meta
        gp = g_poses.float().contiguous() if g_poses is not None else None
        gq = g_patches.float().contiguous() if g_patches is not None else None
        o_p, o_q = torch.empty_like(p7), torch.empty_like(q)
        o_dX, o_dZ = torch.zeros_like(dXc), torch.empty_like(dZc)
        L.check(L.lib().devo_ba_apply_step_backward(L.ptr(p7), L.ptr(q), L.ptr(dXc) if n_opt > 0 else None, L.ptr(dZc), L.ptr(gp), L.ptr(gq), N, Np, P,
                                                    fixedp, n_opt, 1e-3, 10.0, L.ptr(o_p), L.ptr(o_q), L.ptr(o_dX) if n_opt > 0 else None, L.ptr(o_dZ),
                                                    L.stream()), "ba.apply_step_backward")
        return o_p, o_q, o_dX, o_dZ, None, None


def BA(poses, patches, intrinsics, targets, weights, lmbda, ii, jj, kk, bounds, ep=100.0, PRINT=False,
       fixedp=1, structure_only=False, n_frames=None):
    """One differentiable Gauss-Newton step (devo/ba.py:86-182).  `n_frames` (optional, = max(ii, jj) + 1) saves the
    one host synchronisation this function otherwise needs to size the pose system."""
    dev, dt = patches.device, patches.dtype
    if n_frames is None:
        n_frames = int(torch.maximum(ii.max(), jj.max())) + 1
    n = int(n_frames) - fixedp
    coords, ok, (Ji, Jj, Jz) = pops.transform(poses, patches, intrinsics, ii, jj, kk, jacobian=True)
    if _fused_path(patches, lmbda, max(n, 0)) and not PRINT and coords.shape[-1] == 2 and targets.dtype == torch.float32:
        # fp32 on the GPU: edge terms, system + solve and their adjoints in HIP (DEVO_BA_TORCH=1 keeps the torch composition below)
        E, n_opt = ii.numel(), (0 if structure_only else max(n, 0))
        terms = _EdgeTerms.apply(coords, ok, Ji, Jj, Jz, targets, weights, [float(b) for b in bounds])
        return _fused_step(poses, patches, terms, lmbda, ii, jj, kk, fixedp, n_opt, ep)
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

    if _fused_path(patches, lmbda, max(n, 0)) and not (structure_only and False):
        # fp32 on the GPU: system + solve + adjoint in HIP (DEVO_BA_TORCH=1 keeps the torch composition below)
        E, n_opt = ii.numel(), (0 if structure_only else max(n, 0))
        terms = torch.cat([r, w, Jz, (-Ji).reshape(E, 12), Jj.reshape(E, 12)], dim=1)      # Ji enters the kernels as -d coords / d xi_i
        return _fused_step(poses, patches, terms, lmbda, ii, jj, kk, fixedp, n_opt, ep)

    # Patches: the reference compacts to the patches that have edges (torch.unique, ba.py:104).  Working on ALL patch
    # slots instead gives the same update (a slot without edges has C = u = 0 and a zero column of E, hence dZ = 0) and
    # needs neither the data-dependent output size nor the host synchronisation of unique().  Only a per-patch damping
    # vector, which is indexed by the compacted list, still goes through unique().
    per_patch_lmbda = isinstance(lmbda, torch.Tensor) and lmbda.numel() > 1
    if per_patch_lmbda:
        kx, ku = torch.unique(kk, return_inverse=True, sorted=True)
        m = kx.shape[0]
        lmbda = lmbda.reshape(m)
    else:
        kx, ku, m = None, kk, patches.shape[1]
    a, b_ = ii - fixedp, jj - fixedp
    six = torch.arange(6, device=dev)
    n6 = 6 * max(n, 0)
    # flat accumulators: index_add (atomics) instead of index_put(accumulate=True), which sorts its indices on the GPU
    S = torch.zeros(n6 * n6, dtype=dt, device=dev)
    Emat = torch.zeros(n6 * m, dtype=dt, device=dev)
    v = torch.zeros(n6, dtype=dt, device=dev)
    if n > 0:
        # frames outside the optimised window [fixedp, fixedp + n) contribute nothing: their blocks are zeroed and their
        # (clamped) indices point at block 0 — no boolean-mask indexing, no host synchronisation
        ina, inb = ((a >= 0) & (a < n)), ((b_ >= 0) & (b_ < n))
        ca, cb = a.clamp(0, n - 1), b_.clamp(0, n - 1)
        fa, fb = ina.to(dt), inb.to(dt)
        for (ra, ma, Ja), (rb, mb, Jb) in (((ca, fa, Ji), (ca, fa, Ji)), ((ca, fa, Ji), (cb, fb, Jj)),
                                           ((cb, fb, Jj), (ca, fa, Ji)), ((cb, fb, Jj), (cb, fb, Jj))):
            # sum_r w_r Ja[r,p] Jb[r,q] as broadcast products (an einsum here becomes 18 000 batched 6x2x6 GEMMs: 140 us)
            wj = (w * (ma * mb)[:, None])[:, :, None] * Ja
            blk = wj[:, 0, :, None] * Jb[:, 0, None, :] + wj[:, 1, :, None] * Jb[:, 1, None, :]
            rows = (6 * ra)[:, None, None] + six[None, :, None]
            cols = (6 * rb)[:, None, None] + six[None, None, :]
            S = S.index_add(0, (rows * n6 + cols).reshape(-1), blk.reshape(-1))
        for ra, ma, Ja in ((ca, fa, Ji), (cb, fb, Jj)):
            rows = (6 * ra)[:, None] + six[None]
            wm = w * ma[:, None]
            Emat = Emat.index_add(0, (rows * m + ku[:, None]).reshape(-1), (((wm * Jz)[:, :, None] * Ja).sum(1)).reshape(-1))
            v = v.index_add(0, rows.reshape(-1), (((wm * r)[:, :, None] * Ja).sum(1)).reshape(-1))
    S, Emat = S.view(n6, n6), Emat.view(n6, m)
    C = torch.zeros(m, dtype=dt, device=dev).index_add(0, ku, (w * Jz * Jz).sum(-1))
    u = torch.zeros(m, dtype=dt, device=dev).index_add(0, ku, (w * Jz * r).sum(-1))
    # patch slots without an edge have C = 0: with lmbda == 0 (or one that underflows) 1 / (C + lmbda) would be inf and
    # inf * 0 = NaN would poison every depth; the reference only ever sees observed patches (torch.unique, ba.py:113-118)
    Q = torch.where(C + lmbda > 0, 1.0 / (C + lmbda), torch.zeros_like(C))

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
    if kx is None:
        disp = patches[:, :, 2] + dZ.view(1, m, 1, 1)
    else:
        disp = patches[:, :, 2] + torch.zeros_like(patches[:, :, 2]).index_add(1, kx, dZ.view(1, m, 1, 1).expand(1, m, P, P))
    patches = torch.stack([patches[:, :, 0], patches[:, :, 1], disp.clamp(min=1e-3, max=10.0)], dim=2)
    if dX is not None:
        upd = torch.zeros(1, poses.data.shape[1], 6, dtype=dt, device=dev)
        upd[:, fixedp:fixedp + n] = dX.view(1, n, 6)
        poses = poses.retr(upd)
    return poses, patches
