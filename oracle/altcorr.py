"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of the reference's `cuda_corr` extension (altcorr):
  corr forward   devo/altcorr/correlation_kernel.cu:82-136 (kernel) + :193-233 (bilinear epilogue, permute)
  corr backward  :139-190 (kernel) + :236-286 (grad expansion prologue)
  patchify fwd   :16-47 + :288-307 ;  patchify bwd :49-80 + :310-333
  Python wrapper devo/altcorr/correlation.py:51-72

The reference ships no CPU implementation and no golden vectors for these ops and the CUDA
source cannot be built here (THC/THCAtomics.cuh is gone from torch 2.10; no GPU):
"parity unpinned" by reference output — pinned by closed-form known-answer tests
(tests/test_oracle_altcorr.py) and by an independent scalar-loop restatement below.

All functions take/return torch CPU tensors; `acc` selects the accumulation dtype (the fp64
result is the ground truth the HIP kernels are compared with).
"""
import torch


def _floor_int(v):
    """static_cast<int>(floor(v))  (correlation_kernel.cu:118-119)"""
    return torch.floor(v).to(torch.int64)


def corr_raw(fmap1, fmap2, coords, ii, jj, radius, acc=torch.float64, chunk=96):
    """corr_forward_kernel: raw[b,e,a,c,i0,j0] (a = row/y offset, c = col/x offset)."""
    B, E = coords.shape[:2]
    H, W = coords.shape[3:]
    C = fmap1.shape[2]
    H2, W2 = fmap2.shape[3:]
    R, D = radius, 2 * radius + 2
    out = torch.zeros(B, E, D, D, H, W, dtype=acc)
    f2t = fmap2.to(acc).permute(0, 1, 3, 4, 2).reshape(B, fmap2.shape[1], H2 * W2, C)
    f1 = fmap1.to(acc)                                    # [B, Np, C, H, W]
    offs = torch.arange(D) - R
    for b in range(B):
        for e0 in range(0, E, chunk):
            e1 = min(E, e0 + chunk)
            x = coords[b, e0:e1, 0].float()               # [e,H,W]
            y = coords[b, e0:e1, 1].float()
            i1 = _floor_int(y)[:, None, None] + offs[None, :, None, None, None]   # [e,D,1,H,W]
            j1 = _floor_int(x)[:, None, None] + offs[None, None, :, None, None]   # [e,1,D,H,W]
            i1, j1 = torch.broadcast_tensors(i1, j1)
            ok = (i1 >= 0) & (i1 < H2) & (j1 >= 0) & (j1 < W2)
            lin = (i1.clamp(0, H2 - 1) * W2 + j1.clamp(0, W2 - 1))                 # [e,D,D,H,W]
            g2 = f2t[b][jj[e0:e1, None], lin.reshape(e1 - e0, -1)]                 # [e,D*D*H*W,C]
            g2 = g2.view(e1 - e0, D, D, H, W, C)
            g1 = f1[b][ii[e0:e1]].permute(0, 2, 3, 1)                              # [e,H,W,C]
            s = (g2 * g1[:, None, None]).sum(-1)
            out[b, e0:e1] = torch.where(ok, s, torch.zeros_like(s))
    return out


def corr_forward(fmap1, fmap2, coords, ii, jj, radius, acc=torch.float64):
    """cuda_corr.forward: returns the permuted [B,E,c,a,H,W] tensor (contiguous here)."""
    D = 2 * radius + 2
    raw = corr_raw(fmap1, fmap2, coords, ii, jj, radius, acc)
    x = coords[:, :, 0, None, None].float()
    y = coords[:, :, 1, None, None].float()
    dx = (x - x.floor()).to(acc)           # fp32 subtraction first (correlation_kernel.cu:223-224)
    dy = (y - y.floor()).to(acc)
    out = (1 - dx) * (1 - dy) * raw[:, :, 0:D - 1, 0:D - 1]
    out = out + dx * (1 - dy) * raw[:, :, 0:D - 1, 1:D]
    out = out + (1 - dx) * dy * raw[:, :, 1:D, 0:D - 1]
    out = out + dx * dy * raw[:, :, 1:D, 1:D]
    return out.permute(0, 1, 3, 2, 4, 5).contiguous()


def corr_backward(fmap1, fmap2, coords, ii, jj, grad, radius, acc=torch.float64):
    """cuda_corr.backward -> (fmap1_grad, fmap2_grad); no gradient for coords."""
    B, E = coords.shape[:2]
    H, W = coords.shape[3:]
    C = fmap1.shape[2]
    H2, W2 = fmap2.shape[3:]
    R, D = radius, 2 * radius + 2
    g = grad.to(acc).permute(0, 1, 3, 2, 4, 5)                 # back to [B,E,a,c,H,W]
    x = coords[:, :, 0, None, None].float()
    y = coords[:, :, 1, None, None].float()
    dx = (x - x.floor()).to(acc)
    dy = (y - y.floor()).to(acc)
    cg = torch.zeros(B, E, D, D, H, W, dtype=acc)
    cg[:, :, 0:D - 1, 0:D - 1] += (1 - dx) * (1 - dy) * g
    cg[:, :, 0:D - 1, 1:D] += dx * (1 - dy) * g
    cg[:, :, 1:D, 0:D - 1] += (1 - dx) * dy * g
    cg[:, :, 1:D, 1:D] += dx * dy * g
    d1 = torch.zeros(fmap1.shape, dtype=acc)
    d2 = torch.zeros(fmap2.shape, dtype=acc)
    d2f = d2.view(B, fmap2.shape[1], C, H2 * W2)
    f1 = fmap1.to(acc)
    f2f = fmap2.to(acc).reshape(B, fmap2.shape[1], C, H2 * W2)
    offs = torch.arange(D) - R
    for b in range(B):
        for e in range(E):
            i1 = _floor_int(coords[b, e, 1].float())[None, None] + offs[:, None, None, None]
            j1 = _floor_int(coords[b, e, 0].float())[None, None] + offs[None, :, None, None]
            i1, j1 = torch.broadcast_tensors(i1, j1)            # [D,D,H,W]
            ok = (i1 >= 0) & (i1 < H2) & (j1 >= 0) & (j1 < W2)
            lin = (i1.clamp(0, H2 - 1) * W2 + j1.clamp(0, W2 - 1))
            gg = torch.where(ok, cg[b, e], torch.zeros_like(cg[b, e]))             # [D,D,H,W]
            p1 = f1[b, ii[e]]                                                       # [C,H,W]
            win = f2f[b, jj[e]][:, lin.reshape(-1)].view(C, D, D, H, W)
            d1[b, ii[e]] += (gg[None] * win).sum((1, 2))
            contrib = (gg[None] * p1[:, None, None]).reshape(C, -1)                 # [C,D*D*H*W]
            d2f[b, jj[e]].index_add_(1, lin.reshape(-1), contrib)
    return d1.to(fmap1.dtype), d2.to(fmap2.dtype)


def corr_forward_scalar(fmap1, fmap2, coords, ii, jj, radius):
    """Independent pure-Python-loop restatement (tiny cases only): used to cross-check the
    vectorised functions above, one thread of correlation_kernel.cu:102-135 at a time."""
    import math
    B, E = coords.shape[:2]
    H, W = coords.shape[3:]
    C = fmap1.shape[2]
    H2, W2 = fmap2.shape[3:]
    R, D = radius, 2 * radius + 2
    raw = torch.zeros(B, E, D, D, H, W, dtype=torch.float64)
    for b in range(B):
        for e in range(E):
            for i0 in range(H):
                for j0 in range(W):
                    x = float(coords[b, e, 0, i0, j0]); y = float(coords[b, e, 1, i0, j0])
                    for a in range(D):
                        for c in range(D):
                            i1 = int(math.floor(y)) + a - R
                            j1 = int(math.floor(x)) + c - R
                            if 0 <= i1 < H2 and 0 <= j1 < W2:
                                raw[b, e, a, c, i0, j0] = float(
                                    (fmap1[b, ii[e], :, i0, j0].double() * fmap2[b, jj[e], :, i1, j1].double()).sum())
    out = torch.zeros(B, E, D - 1, D - 1, H, W, dtype=torch.float64)
    for b in range(B):
        for e in range(E):
            for i0 in range(H):
                for j0 in range(W):
                    x = coords[b, e, 0, i0, j0].float(); y = coords[b, e, 1, i0, j0].float()
                    dx = float(x - x.floor()); dy = float(y - y.floor())
                    r = raw[b, e, :, :, i0, j0]
                    out[b, e, :, :, i0, j0] = (1 - dx) * (1 - dy) * r[:-1, :-1] + dx * (1 - dy) * r[:-1, 1:] \
                        + (1 - dx) * dy * r[1:, :-1] + dx * dy * r[1:, 1:]
    return out.permute(0, 1, 3, 2, 4, 5).contiguous()


def patchify_forward(net, coords, radius):
    """cuda_corr.patchify_forward: patches[b,m,k,a,c] = net[b,k,floor(y)+a-R,floor(x)+c-R] or 0."""
    B, M = coords.shape[:2]
    C, H, W = net.shape[1:]
    R, D = radius, 2 * radius + 2
    offs = torch.arange(D) - R
    i = _floor_int(coords[..., 1].float())[:, :, None, None] + offs[None, None, :, None]
    j = _floor_int(coords[..., 0].float())[:, :, None, None] + offs[None, None, None, :]
    i, j = torch.broadcast_tensors(i, j)                          # [B,M,D,D]
    ok = (i >= 0) & (i < H) & (j >= 0) & (j < W)
    lin = (i.clamp(0, H - 1) * W + j.clamp(0, W - 1)).view(B, -1)  # [B,M*D*D]
    g = torch.gather(net.reshape(B, C, H * W), 2, lin[:, None].expand(B, C, lin.shape[1]))
    g = g.view(B, C, M, D, D).permute(0, 2, 1, 3, 4)
    return torch.where(ok[:, :, None], g, torch.zeros_like(g)).contiguous()


def patchify_backward(net, coords, grad, radius):
    """cuda_corr.patchify_backward: scatter-add of patch gradients into a zero net gradient."""
    B, M = coords.shape[:2]
    C, H, W = net.shape[1:]
    R, D = radius, 2 * radius + 2
    offs = torch.arange(D) - R
    i = _floor_int(coords[..., 1].float())[:, :, None, None] + offs[None, None, :, None]
    j = _floor_int(coords[..., 0].float())[:, :, None, None] + offs[None, None, None, :]
    i, j = torch.broadcast_tensors(i, j)
    ok = (i >= 0) & (i < H) & (j >= 0) & (j < W)
    lin = (i.clamp(0, H - 1) * W + j.clamp(0, W - 1)).view(B, -1)
    g = torch.where(ok[:, :, None], grad, torch.zeros_like(grad))          # [B,M,C,D,D]
    g = g.permute(0, 2, 1, 3, 4).reshape(B, C, -1).double()
    out = torch.zeros(B, C, H * W, dtype=torch.float64)
    out.scatter_add_(2, lin[:, None].expand(B, C, lin.shape[1]), g)
    return out.view(B, C, H, W).to(net.dtype)


def patchify(net, coords, radius, mode="bilinear"):
    """devo/altcorr/correlation.py:51-68"""
    patches = patchify_forward(net, coords, radius)
    if mode == "bilinear":
        offset = coords - coords.floor()
        dx, dy = offset[:, :, None, None, None].unbind(dim=-1)
        d = 2 * radius + 1
        x00 = (1 - dy) * (1 - dx) * patches[..., :d, :d]
        x01 = (1 - dy) * dx * patches[..., :d, 1:]
        x10 = dy * (1 - dx) * patches[..., 1:, :d]
        x11 = dy * dx * patches[..., 1:, 1:]
        return x00 + x01 + x10 + x11
    return patches
