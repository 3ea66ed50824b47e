"""Host-side mirror of devo/projective_ops.py (iproj :19-29, proj :32-50, transform :53-105,
point_cloud :107-109, flow_mag :111-121).

Two execution paths with identical results:
  * no gradient needed (inference, DEVO.update -> devo.py:222): ONE fused HIP kernel (cuda_ba.transform);
  * autograd needed (training, enet.py:341,363-369): the same maths as a torch composition over the HIP
    SE3 ops of devo_amd.lietorch, so gradients flow exactly as in the reference.
"""
import torch
from .lietorch import SE3
from .backends import cuda_ba

MIN_DEPTH = 0.2


def _K(intrinsics):
    return [intrinsics[..., k, None, None] for k in range(4)]


def iproj(patches, intrinsics):
    fx, fy, cx, cy = _K(intrinsics)
    px, py, pd = patches[:, :, 0], patches[:, :, 1], patches[:, :, 2]
    return torch.stack([(px - cx) / fx, (py - cy) / fy, torch.ones_like(pd), pd], dim=-1)


def proj(Xh, intrinsics, depth=False):
    fx, fy, cx, cy = _K(intrinsics)
    rz = 1.0 / Xh[..., 2].clamp(min=0.1)
    u = fx * (rz * Xh[..., 0]) + cx
    v = fy * (rz * Xh[..., 1]) + cy
    return torch.stack([u, v, rz] if depth else [u, v], dim=-1)


def _needs_grad(*ts):
    return torch.is_grad_enabled() and any(t.requires_grad for t in ts)


class _FusedTransform(torch.autograd.Function):
    """The fused reprojection kernel (devo_transform) with its adjoint (devo_transform_vjp): gradients reach poses (lietorch's
    6-of-7 convention) and patches through the coordinates AND through the Jacobians Ji, Jj, Jz (second-order terms), which
    is what the chained Gauss-Newton steps of training need (enet.py:353-369).  Intrinsics get no gradient."""

    @staticmethod
    def forward(ctx, pose_data, patches, intrinsics, ii, jj, kk, depth, jacobian, tonly):
        out = cuda_ba.transform(pose_data, patches, intrinsics, ii, jj, kk, depth=depth, valid=True, jacobian=jacobian, tonly=tonly,
                                layout="pp2")
        ctx.save_for_backward(pose_data, patches, intrinsics, ii, jj, kk)
        ctx.flags = (depth, jacobian, tonly)
        if jacobian:
            c, v, (Ji, Jj, Jz) = out
            ctx.mark_non_differentiable(v)
            return c, v, Ji, Jj, Jz
        c, v = out
        ctx.mark_non_differentiable(v)
        return c, v

    @staticmethod
    def backward(ctx, g_c, g_v, g_Ji=None, g_Jj=None, g_Jz=None):
        pose_data, patches, intrinsics, ii, jj, kk = ctx.saved_tensors
        depth, jacobian, tonly = ctx.flags
        gJ = (g_Ji, g_Jj, g_Jz) if jacobian else None
        gp, gq = cuda_ba.transform_vjp(pose_data, patches, intrinsics, ii, jj, kk, g_c, gJ, depth=depth, tonly=tonly)
        return gp.view_as(pose_data), gq.view_as(patches), None, None, None, None, None, None, None


_VIEW_2PP = __import__("os").environ.get("DEVO_TRANSFORM_2PP_VIEW", "1") != "0"     # 0: the no-gradient coordinates as a contiguous [1,E,P,P,2] tensor


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False, tonly=False, fused=True):
    """coords [1,E,P,P,2(+1)] (+ validity [1,E], + (Ji [1,E,2,6], Jj [1,E,2,6], Jz [1,E,2,1])).
    fused=False: always the reference's composition over the SE3 group ops (projective_ops.py:53-105: what an unmodified checkout of the
    reference runs on top of the installed lietorch_backends; bench.py --api reference times it)."""
    import os
    fused_ok = fused and poses.data.dtype == torch.float32 and poses.data.shape[0] == 1 and patches.is_cuda
    if fused_ok and not _needs_grad(poses.data, patches, intrinsics):
        if _VIEW_2PP and not (depth or valid or jacobian):
            # coordinates only (devo.py:222, :352, flow_mag): every caller of the reference turns [1,E,P,P,2] into [1,E,2,P,P] right away
            # (`.permute(0, 1, 4, 2, 3).contiguous()`, devo.py:223) or reads it through arithmetic — so the kernel writes the 2 x P x P layout and
            # the result is handed out as the [1,E,P,P,2] VIEW of it: the caller's permute + contiguous() finds contiguous memory and copies nothing
            c = cuda_ba.transform(poses.data, patches, intrinsics, ii, jj, kk, tonly=tonly, layout="2pp")
            return c.permute(0, 1, 3, 4, 2)
        return cuda_ba.transform(poses.data, patches, intrinsics, ii, jj, kk, depth=depth, valid=valid,
                                 jacobian=jacobian, tonly=tonly, layout="pp2")
    # (not for tonly: the reference's autograd treats the overwritten quaternion slots of `Gij.data[..., 3:] = identity` as cut
    #  — projective_ops.py:62 — which is not the derivative of the function; the composition below reproduces that, and only
    #  flow_mag, which needs no gradient, uses tonly)
    if (fused_ok and not tonly and not _needs_grad(intrinsics) and patches.dtype == torch.float32 and patches.shape[-1] ** 2 <= 25
            and os.environ.get("DEVO_TRANSFORM_TORCH", "0") != "1"):
        # fp32 on the GPU with gradients: one forward kernel, one adjoint kernel (DEVO_TRANSFORM_TORCH=1: the composition below)
        out = _FusedTransform.apply(poses.data, patches, intrinsics, ii, jj, kk, bool(depth), bool(jacobian), bool(tonly))
        if jacobian:
            c, v, Ji, Jj, Jz = out
            return c, v, (Ji, Jj, Jz)
        return (out[0], out[1]) if valid else out[0]

    # gathers along the edge axis with index_select: its backward is an atomic index_add, whereas the backward of
    # advanced indexing (`x[:, idx]`) sorts the indices on the GPU (0.3 ms per gather at 18 000 edges)
    SE3 = type(poses)
    take = lambda t, idx: torch.index_select(t, 1, idx)
    Gij = SE3(take(poses.data, jj)) * SE3(take(poses.data, ii)).inv()
    if tonly:
        Gij.data[..., 3:] = torch.eye(4, dtype=Gij.data.dtype, device=Gij.data.device)[3]      # (made on the device: no blocking host -> device copy)
    X1 = Gij[:, :, None, None] * iproj(take(patches, kk), take(intrinsics, ii))
    c = X1.shape[2] // 2
    intr_j = take(intrinsics, jj)
    x1 = proj(X1, intr_j, depth)
    if jacobian:
        X, Y, Z, H = X1[..., c, c, :].unbind(dim=-1)
        fx, fy = intr_j[..., 0], intr_j[..., 1]
        o = torch.zeros_like(Z)
        big = Z.abs() > 0.2
        d = torch.where(big, 1.0 / torch.where(big, Z, torch.ones_like(Z)), o)
        Jj = torch.stack([
            torch.stack([fx * d * H, o, -fx * X * d * d * H, -fx * X * d * d * Y,
                         fx * d * Z + fx * X * d * d * X, -fx * d * Y], -1),
            torch.stack([o, fy * d * H, -fy * Y * d * d * H, -fy * d * Z - fy * Y * d * d * Y,
                         fy * Y * d * d * X, fy * d * X], -1)], dim=-2)
        Ji = -Gij[:, :, None].adjT(Jj)
        # (the translation as the reference reads it, projective_ops.py:97: the last column of Gij.matrix(), i.e. through the group ACTION — its
        #  gradient then reaches Gij in lietorch's tangent convention, rotation part included; slicing Gij.data instead has the same value and a
        #  different (Euclidean, translation-only) gradient, which two chained BA steps expose: tests/test_gpu_train_iteration.py)
        t = Gij.matrix()[..., :3, 3]
        Jz = torch.stack([fx * d * t[..., 0] - fx * X * d * d * t[..., 2],
                          fy * d * t[..., 1] - fy * Y * d * d * t[..., 2]], -1)[..., None]
        return x1, (Z > 0.2).float(), (Ji, Jj, Jz)
    if valid:
        return x1, (X1[..., c, c, 2] > 0.2).float()
    return x1


def point_cloud(poses, patches, intrinsics, ix):
    return poses[:, ix, None, None].inv() * iproj(patches, intrinsics[:, ix])


def flow_mag(poses, patches, intrinsics, ii, jj, kk, beta=0.3):
    c0 = transform(poses, patches, intrinsics, ii, ii, kk)
    full = (transform(poses, patches, intrinsics, ii, jj, kk) - c0).norm(dim=-1)
    trans = (transform(poses, patches, intrinsics, ii, jj, kk, tonly=True) - c0).norm(dim=-1)
    return beta * full + (1 - beta) * trans
