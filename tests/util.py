import torch


def rel_err(a, b):
    """max |a-b| relative to the scale of the reference tensor b."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))


def assert_rel(a, b, tol, what=""):
    err = rel_err(a, b)
    assert err <= tol, f"{what}: relative error {err:.3e} > {tol:.1e}"


def channels_last5(x):
    """[B,n,C,H,W] logical tensor stored as [B,n,H,W,C]."""
    return x.permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)


def row_rel_err(a, b, floor_frac=1e-2):
    """Per-ROW relative error: max over rows of |a - b|_inf / max(|b_row|_inf, floor_frac * |b|_inf) — a small pose component or
    a small inverse depth is held to 1e-4 of ITSELF (down to `floor_frac` of the tensor's scale), unlike rel_err, which scales
    every element by the tensor's maximum.  Rows = the last dimension (a 1-D tensor is a column of scalars)."""
    a, b = a.double().cpu(), b.double().cpu()
    if a.dim() == 1:
        a, b = a[:, None], b[:, None]
    a, b = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    scale = b.abs().amax(dim=1).clamp(min=floor_frac * float(b.abs().max().clamp(min=1e-30)))
    return (a - b).abs().amax(dim=1) / scale


def shuffled_plan(plan, n_slots, seed=0):
    """A lookup plan (cuda_corr.plan: n slots | #heavy | n scratch | #dead) with the heavy slots, the live slots and the dead slots
    shuffled among themselves: the same classes, different neighbours.  The lookup's results must not change by one bit."""
    p = plan.clone().cpu()
    nh, nd = int(p[n_slots]), int(p[2 * n_slots + 1])
    g = torch.Generator().manual_seed(seed)
    for lo, hi in ((0, nh), (nh, n_slots - nd), (n_slots - nd, n_slots)):
        if hi > lo:
            p[lo:hi] = p[lo:hi][torch.randperm(hi - lo, generator=g)]
    return p.to(plan.device)
