"""Host-side mirror of devo/altcorr/correlation.py (CorrLayer :5-31, PatchLayer :34-49, patchify :51-68,
corr :71-72) over the HIP kernels, plus the fused two-level lookup used by DEVO.corr / CorrBlock."""
import torch
from .backends import cuda_corr


DROPOUT_SIGMAS = 6.0         # capacity of an edge subset: its expected size + this many standard deviations of the binomial count


def _edge_subset(n_edges, dropout, device):
    """correlation.py:20-25 draws `torch.rand(len(ii)) < dropout` and indexes with the boolean mask — a host synchronisation for the subset's
    size, twice per update iteration of a training step.  Here the SAME Bernoulli draw is compacted into a list of FIXED capacity (expected
    size + 6 sigma of the binomial count: 3 930 slots for 18 000 edges at 0.2, 9 % more than the 3 600 expected) without leaving the device:
    a stable sort of the mask puts the kept edges first (in edge order), `valid` marks the slots that hold one; the others carry OTHER
    (distinct, not drawn) edges with a zero gradient — spread over the frames like real ones, where copies of one edge would pile their
    windows onto one tile of the backward's frame kernel.  (A draw beyond the capacity — probability ~1e-9 — would drop its last edges.)
    Returns (keep [K] int64, valid [K] bool)."""
    mask = torch.rand(n_edges, device=device) < dropout
    cap = min(n_edges, int(n_edges * dropout + DROPOUT_SIGMAS * (n_edges * dropout * (1.0 - dropout)) ** 0.5) + 1)
    order = torch.sort((~mask).to(torch.uint8), stable=True).indices[:cap]
    valid = mask[order]
    return order, valid


class CorrLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap1, fmap2, coords, ii, jj, radius, dropout):
        ctx.save_for_backward(fmap1, fmap2, coords, ii, jj)
        ctx.radius, ctx.dropout = radius, dropout
        corr, = cuda_corr.forward(fmap1, fmap2, coords, ii, jj, radius)
        return corr

    @staticmethod
    def backward(ctx, grad):
        fmap1, fmap2, coords, ii, jj = ctx.saved_tensors
        if ctx.dropout < 1:
            # correlation.py:20-25: only a random subset of edges propagates gradient; the draw stays here, its compaction on the device
            keep, valid = _edge_subset(len(ii), ctx.dropout, ii.device)
            coords, ii, jj = coords.index_select(1, keep), ii.index_select(0, keep), jj.index_select(0, keep)
            grad = grad.index_select(1, keep) * valid.view(1, -1, *([1] * (grad.dim() - 2))).to(grad.dtype)
        d1, d2 = cuda_corr.backward(fmap1, fmap2, coords, ii, jj, grad, ctx.radius)
        return d1, d2, None, None, None, None, None


class PatchLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, coords, radius):
        ctx.radius = radius
        ctx.save_for_backward(net, coords)
        patches, = cuda_corr.patchify_forward(net, coords, radius)
        return patches

    @staticmethod
    def backward(ctx, grad):
        net, coords = ctx.saved_tensors
        g, = cuda_corr.patchify_backward(net, coords, grad, ctx.radius)
        return g, None, None


def patchify(net, coords, radius, mode="bilinear"):
    """(2r+1)^2 patches of `net` around `coords`, bilinear in the sub-pixel offset (correlation.py:51-68)."""
    patches = PatchLayer.apply(net, coords, radius)
    if mode != "bilinear":
        return patches
    frac = (coords - coords.floor()).to(net.device)
    dx, dy = frac[:, :, None, None, None].unbind(dim=-1)
    d = 2 * radius + 1
    return ((1 - dy) * (1 - dx) * patches[..., :d, :d] + (1 - dy) * dx * patches[..., :d, 1:]
            + dy * (1 - dx) * patches[..., 1:, :d] + dy * dx * patches[..., 1:, 1:])


def corr(fmap1, fmap2, coords, ii, jj, radius=1, dropout=1):
    return CorrLayer.apply(fmap1, fmap2, coords, ii, jj, radius, dropout)


class CorrPyramidLayer(torch.autograd.Function):
    """torch.stack([corr(fmap1, pyramid[l], coords / scales[l], ii, jj, radius, dropout) for l], -1).view(1, E, -1) (enet.py:203-216) with
    the forward as ONE fused launch writing the stacked layout (no per-level tensors, no scaled coordinates, no stack copy) and the
    backward as the per-level kernels on the strided halves of the incoming gradient; every level draws its own edge subset, as the
    reference's two CorrLayer.backward calls do (correlation.py:20-25)."""

    @staticmethod
    def forward(ctx, fmap1, coords, ii, jj, radius, dropout, scales, *pyramid):
        ctx.save_for_backward(fmap1, coords, ii, jj, *pyramid)
        ctx.radius, ctx.dropout, ctx.scales = radius, dropout, tuple(scales)
        return cuda_corr.forward_pyramid(fmap1, list(pyramid), coords, ii, jj, radius, scales)

    @staticmethod
    def backward(ctx, grad):
        fmap1, coords, ii, jj, *pyramid = ctx.saved_tensors
        nl, E = len(pyramid), coords.shape[1]
        D = 2 * ctx.radius + 1
        g = grad.view(grad.shape[0], E, D, D, coords.shape[3], coords.shape[4], nl)
        d1, d2s = None, []
        # every level draws its own edge subset (the reference's two CorrLayer.backward calls do); compacted on the device: no host sync
        keeps = [_edge_subset(len(ii), ctx.dropout, ii.device) for _ in range(nl)] if ctx.dropout < 1 else None
        for l in range(nl):
            c_l, g_l, i_l, j_l = coords / ctx.scales[l], g[..., l], ii, jj
            if keeps is not None:
                k, valid = keeps[l]
                c_l, i_l, j_l = c_l.index_select(1, k), ii.index_select(0, k), jj.index_select(0, k)
                g_l = g_l.index_select(1, k) * valid.view(1, -1, 1, 1, 1, 1).to(g_l.dtype)
            a, b = cuda_corr.backward(fmap1, pyramid[l], c_l, i_l, j_l, g_l, ctx.radius)
            d1 = a if d1 is None else d1 + a
            d2s.append(b)
        return (d1, None, None, None, None, None, None, *d2s)


def corr_pyramid(fmap1, pyramid, coords, ii, jj, radius=3, scales=(1, 4), dropout=1):
    """Fused form of
        torch.stack([corr(fmap1, pyramid[l], coords / scales[l], ii, jj, radius, dropout) for l], -1).view(1, E, -1)
    (devo/devo.py:215-217, enet.py:203-216): the levels written straight into the stacked layout by one launch; differentiable with
    respect to fmap1 and the pyramid levels (CorrPyramidLayer) when gradients are enabled."""
    if torch.is_grad_enabled() and (fmap1.requires_grad or any(f.requires_grad for f in pyramid)):
        return CorrPyramidLayer.apply(fmap1, coords, ii, jj, radius, dropout, tuple(scales), *pyramid)
    return cuda_corr.forward_pyramid(fmap1, pyramid, coords, ii, jj, radius, scales)


def channel_blocked(fmap, cb=8, pad=None):
    """Re-lay a [B,n,C,H,W] pyramid level as channel-blocked storage [B, n, C/cb, H, W, cb]: the cb channels of a
    pixel are contiguous and horizontally adjacent pixels follow each other, so one row of a lookup box is one
    contiguous run of memory (full cache lines) for every channel chunk.  Inference lookup only, fp32 or fp16
    (cuda_corr.forward / corr_pyramid accept the 6-D tensor in place of fmap2).
    pad: elements of padding behind every channel block (the returned tensor is a strided view; DEVO_PLANE_PAD overrides the default 0)."""
    import os
    B, n, C, H, W = fmap.shape
    if C % cb:
        raise RuntimeError(f"channel_blocked: C={C} is not a multiple of {cb}")
    if pad is None:
        pad = int(os.environ.get("DEVO_PLANE_PAD", "0"))
    src = fmap.reshape(B, n, C // cb, cb, H, W).permute(0, 1, 2, 4, 5, 3)
    if pad <= 0:
        return src.contiguous()
    buf = torch.zeros(B, n, C // cb, H * W * cb + pad, dtype=fmap.dtype, device=fmap.device)
    out = buf[..., :H * W * cb].view(B, n, C // cb, H, W, cb)
    out.copy_(src)
    return out


def build_pyramid(fmap, out=None, slot=None):
    """devo/devo.py:526-527 + utils.py:70-79 in one kernel: NCHW frames fmap [B, n, C, H, W] -> the two channel-blocked
    pyramid levels the lookup kernel wants, ([B, n, C/8, H, W, 8], [B, n, C/8, H/4, W/4, 8]) (level 1 = 4x4 mean).
    `out=(l0, l1), slot=k`: write the n frames of `fmap` into frames k.. of existing ring buffers instead."""
    from . import _lib as L
    L.require_gpu(fmap)
    B, n, C, H, W = fmap.shape
    fmap = fmap.contiguous()
    if out is None:
        l0 = torch.empty(B, n, C // 8, H, W, 8, dtype=fmap.dtype, device=fmap.device)
        l1 = torch.empty(B, n, C // 8, H // 4, W // 4, 8, dtype=fmap.dtype, device=fmap.device)
        views = [(fmap[b], l0[b], l1[b]) for b in range(B)]
    else:
        l0, l1 = out
        k = int(slot or 0)
        if l0.shape[2:] != (C // 8, H, W, 8) or l1.shape[2:] != (C // 8, H // 4, W // 4, 8) or not (l0.is_contiguous() and l1.is_contiguous()):
            raise RuntimeError("build_pyramid: ring buffers must be contiguous channel-blocked tensors of matching size")
        views = [(fmap[b], l0[b, k:k + n], l1[b, k:k + n]) for b in range(B)]
    for src, d0, d1 in views:
        rc = L.lib().devo_pyramid_build(L.ptr(src), L.ptr(d0), L.ptr(d1), n, C, H, W, C * H * W, d0.stride(0), d1.stride(0),
                                        L.dtype_code(fmap), L.stream())
        L.check(rc, "altcorr.build_pyramid")
    return l0, l1


def channels_last(fmap):
    """Re-lay a [B,n,C,H,W] feature pyramid level as channels-last storage (strides (.., 1, W*C, C)) while
    keeping its logical shape: this is the layout the LDS-staged lookup kernel wants."""
    return fmap.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
