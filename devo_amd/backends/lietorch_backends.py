"""HIP-backed module with the interface of the reference's `lietorch_backends` extension
(devo/lietorch/src/lietorch.cpp:286-316) for SE3 (group_id 3) — the only group the DEVO hot path uses.
Inputs arrive flattened [batch, dim] and contiguous (lietorch.cpp:7 CHECK_CONTIGUOUS).  No CPU fallback."""
import torch
from .. import _lib as L

SE3_ID = 3


def _chk(group_id, *ts):
    if group_id != SE3_ID:
        raise RuntimeError(f"lietorch_backends (devo_amd): only SE3 (group_id 3) is implemented, got {group_id}")
    L.require_gpu(*ts)
    for t in ts:
        if not t.is_contiguous():
            raise RuntimeError("lietorch_backends: input must be contiguous")      # lietorch.cpp:7
        if t.dtype not in (torch.float32, torch.float64):
            raise RuntimeError(f"lietorch_backends: float32/float64 only, got {t.dtype}")
    d = ts[0].dtype
    for t in ts:
        if t.dtype != d:
            raise RuntimeError("lietorch_backends: mixed dtypes")


def _unary(name, out_dim):
    def f(group_id, X):
        _chk(group_id, X)
        n = X.shape[0]
        out = torch.empty(n, out_dim, dtype=X.dtype, device=X.device)
        L.check(getattr(L.lib(), name)(L.ptr(X), L.ptr(out), n, L.dtype_code(X), L.stream()), name)
        return out
    return f


def _unary_bwd(name, out_dim):
    def f(group_id, grad, X):
        _chk(group_id, grad, X)
        n = X.shape[0]
        out = torch.empty(n, out_dim, dtype=X.dtype, device=X.device)
        L.check(getattr(L.lib(), name)(L.ptr(grad), L.ptr(X), L.ptr(out), n, L.dtype_code(X), L.stream()), name)
        return [out]
    return f


def _binary(name, out_dim):
    def f(group_id, X, y):
        _chk(group_id, X, y)
        n = X.shape[0]
        out = torch.empty(n, out_dim, dtype=X.dtype, device=X.device)
        L.check(getattr(L.lib(), name)(L.ptr(X), L.ptr(y), L.ptr(out), n, L.dtype_code(X), L.stream()), name)
        return out
    return f


def _binary_bwd(name, dy_dim):
    def f(group_id, grad, X, y):
        _chk(group_id, grad, X, y)
        n = X.shape[0]
        dX = torch.empty(n, 7, dtype=X.dtype, device=X.device)
        dy = torch.empty(n, dy_dim, dtype=X.dtype, device=X.device)
        L.check(getattr(L.lib(), name)(L.ptr(grad), L.ptr(X), L.ptr(y), L.ptr(dX), L.ptr(dy), n, L.dtype_code(X),
                                       L.stream()), name)
        return [dX, dy]
    return f


expm, expm_backward = _unary("devo_se3_exp", 7), _unary_bwd("devo_se3_exp_backward", 6)
logm, logm_backward = _unary("devo_se3_log", 6), _unary_bwd("devo_se3_log_backward", 7)
inv, inv_backward = _unary("devo_se3_inv", 7), _unary_bwd("devo_se3_inv_backward", 7)
mul, mul_backward = _binary("devo_se3_mul", 7), _binary_bwd("devo_se3_mul_backward", 7)
adj, adj_backward = _binary("devo_se3_adj", 6), _binary_bwd("devo_se3_adj_backward", 6)
adjT, adjT_backward = _binary("devo_se3_adjT", 6), _binary_bwd("devo_se3_adjT_backward", 6)
act, act_backward = _binary("devo_se3_act", 3), _binary_bwd("devo_se3_act_backward", 3)
act4, act4_backward = _binary("devo_se3_act4", 4), _binary_bwd("devo_se3_act4_backward", 4)
Jinv = _binary("devo_se3_jinv", 6)


def as_matrix(group_id, X):
    _chk(group_id, X)
    n = X.shape[0]
    out = torch.empty(n, 4, 4, dtype=X.dtype, device=X.device)
    L.check(L.lib().devo_se3_as_matrix(L.ptr(X), L.ptr(out), n, L.dtype_code(X), L.stream()), "devo_se3_as_matrix")
    return out


def projector(group_id, X):
    raise NotImplementedError("lietorch_backends.projector (ToVec/FromVec) is outside the DEVO hot path "
                              "(SURVEY.md §2.1 row 3): never reached from devo.py / enet.py / train.py")


# The compiled binding (devo_amd._C.lietorch_backends: the same 19 functions taking torch::Tensor) replaces the ctypes forms above when present.
from . import native as _native_binding
_N = _native_binding()
if _N is not None:
    for _name in ("expm", "logm", "inv", "mul", "adj", "adjT", "act", "act4"):
        globals()[_name] = getattr(_N.lietorch_backends, _name)
        globals()[_name + "_backward"] = getattr(_N.lietorch_backends, _name + "_backward")
    as_matrix, Jinv = _N.lietorch_backends.as_matrix, _N.lietorch_backends.Jinv
