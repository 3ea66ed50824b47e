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
