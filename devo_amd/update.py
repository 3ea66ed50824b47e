"""DEVO's `Update` operator (devo/enet.py:32-99, devo/blocks.py:15-48) on the HIP path.

Same module tree and parameter names as the reference (`corr.0.weight`, `agg_kk.f.bias`, `gru.1.gate.0.weight`, ...), so
a reference checkpoint's `update.*` entries load with `load_state_dict`.  Inference (`torch.no_grad()`): the dense
layers are library GEMMs (`torch.nn.functional.linear` -> hipBLASLt), everything between them runs in the fused
kernels of devo_amd/csrc/update.hip — LayerNorm with the residual sums and the ReLU fused in, masked neighbour
gather, SoftAgg's per-group softmax + weighted sum + expand, gated residual, the two 2-wide heads — with the group
tables built by the bundle adjustment's own index kernels (`cuda_ba.prepare`) and `cuda_ba.neighbors`.
With gradients enabled the same computation runs as a plain torch composition (no torch_scatter needed).
"""
import os
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from .backends import cuda_ba

DIM = 384


class _GradClip(torch.autograd.Function):            # blocks.py:72-81: identity forward; backward NaN -> 0, clamp +-0.01
    @staticmethod
    def forward(ctx, x):
        return x

    @staticmethod
    def backward(ctx, grad):
        return torch.nan_to_num(grad, nan=0.0, posinf=float("inf"), neginf=float("-inf")).clamp(min=-0.01, max=0.01)



_wsplit_cache = {}          # (ptr, version, shape, strides, transposed) -> (weight [kept alive], its split image); LRU
WSPLIT_CACHE_ENTRIES = 96
SPLIT_GEMM = __import__("os").environ.get("DEVO_UPD_SPLIT_GEMM", "1") != "0"     # 0: the library's fp32 GEMMs for every Linear layer


def _split_weight(w, transposed):
    """The B-operand image of csrc/linear.hip for `x @ w.T` (transposed=False) or `g @ w` (True): every fp32 weight as an exact fp16
    hi + lo pair, one 1 KB piece per (K step, column tile).  Cached per version of the weight: a training step's 18 update iterations
    split each layer twice (forward, dX), not 36 times; the optimiser's in-place step bumps the version.  (The cache holds the weight, so its
    storage cannot be handed to another tensor; an edit through `.data`, which has its own version counter, is NOT seen — like Update._cat's
    concatenated weights: assign parameters with copy_ / load_state_dict / optimiser steps.)"""
    key = (w.data_ptr(), w._version, tuple(w.shape), tuple(w.stride()), transposed)
    hit = _wsplit_cache.pop(key, None)
    if hit is not None:
        _wsplit_cache[key] = hit
        return hit[1]
    for k in [k for k in _wsplit_cache if k[0] == key[0] and k[4] == transposed]:
        del _wsplit_cache[k]
    while len(_wsplit_cache) >= WSPLIT_CACHE_ENTRIES:
        del _wsplit_cache[next(iter(_wsplit_cache))]
    out_f, in_f = w.shape
    N, K = (in_f, out_f) if transposed else (out_f, in_f)
    s_n, s_k = (w.stride(1), w.stride(0)) if transposed else (w.stride(0), w.stride(1))
    if w.dtype == torch.float16:                                       # fp16 storage: the operand image only (no split, no scales)
        img = torch.empty(N * ((K + 63) // 64 * 64), dtype=torch.float16, device=w.device)
        L.check(L.lib().devo_upd_pack_weight_f16(L.ptr(w), s_n, s_k, N, K, L.ptr(img), L.stream()), "update.pack_weight_f16")
    else:
        Np = (N + 95) // 96 * 96                                        # (whole column blocks of 96: zero columns behind N)
        img = torch.empty(int(L.lib().devo_upd_split_weight_bytes(N, K)) // 4, dtype=torch.float32, device=w.device)
        assert img.numel() == Np * ((K + 31) // 32 * 32) + Np
        L.check(L.lib().devo_upd_split_weight(L.ptr(w), s_n, s_k, N, K, L.ptr(img), L.stream()), "update.split_weight")
    _wsplit_cache[key] = (w, img)
    return img


def invalidate_weight_images():
    """Forget every cached weight image (split fp32 / packed fp16 operands, csrc/linear.hip).  The cache follows a weight's VERSION counter:
    copy_, load_state_dict and optimiser steps are seen; an edit through `.data` (p.data.add_, EMA swaps, old-style optimisers) has its own
    counter and is NOT — call this (or Update.invalidate_weights) after such an edit.  Update.train() / .eval() / load_state_dict call it."""
    _wsplit_cache.clear()


def _split_ok(x2, n_out, k_in):
    return (SPLIT_GEMM and x2.is_cuda and x2.dtype == torch.float32 and x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) >= k_in
            and n_out >= 96 and k_in >= 32 and x2.shape[0] >= 1024 and      # (the 2-wide heads: the library's 10 us against 11-18)
            ((x2.shape[0] - 1) * x2.stride(0) + k_in) * 4 < (1 << 31))


F16_GEMM = __import__("os").environ.get("DEVO_UPD_F16_GEMM", "1") != "0"          # 0: the library's GEMMs for the fp16 operator


def _f16_ok(x2, n_out, k_in):
    return (F16_GEMM and x2.is_cuda and x2.dtype == torch.float16 and x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) >= k_in
            and x2.stride(0) % 2 == 0 and x2.data_ptr() % 4 == 0 and n_out % 96 == 0 and x2.shape[0] >= 1024
            and ((x2.shape[0] - 1) * x2.stride(0) + k_in) * 2 < (1 << 31))


RS_GEMM = __import__("os").environ.get("DEVO_UPD_RS", "1") != "0"                # 0: every fp16 Linear layer on csrc/linear.hip's kernel
RS_CHAINS = __import__("os").environ.get("DEVO_UPD_RS_CHAINS", "1") != "0"       # 0: no row-resident chains (one launch per layer / LayerNorm)


def _rs_image(w):
    """The B-operand image of csrc/gemm_rs.hip for a [384 n, K] fp16 weight, cached per version like _split_weight's images."""
    key = (w.data_ptr(), w._version, tuple(w.shape), tuple(w.stride()), "rs")
    hit = _wsplit_cache.pop(key, None)
    if hit is not None:
        _wsplit_cache[key] = hit
        return hit[1]
    for k in [k for k in _wsplit_cache if k[0] == key[0] and k[4] == "rs"]:
        del _wsplit_cache[k]
    while len(_wsplit_cache) >= WSPLIT_CACHE_ENTRIES:
        del _wsplit_cache[next(iter(_wsplit_cache))]
    N, K = w.shape
    img = torch.empty(int(L.lib().devo_upd_rs_weight_bytes(N, K)) // 2, dtype=torch.float16, device=w.device)
    L.check(L.lib().devo_upd_rs_pack_weight_f16(L.ptr(w), w.stride(0), w.stride(1), N, K, L.ptr(img), L.stream()), "update.rs_pack_weight")
    _wsplit_cache[key] = (w, img)
    return img


def _linear_f16(x2, w, b, relu=False, relu_from=None, residual=None, out=None):
    """x2 [rows, K] fp16 -> act(x2 @ w.T + b) [+ residual] in fp16 storage with fp32 accumulation (csrc/gemm_rs.hip: rows in LDS once, weights
    from the L2 into registers — the 384-wide layers; csrc/linear.hip: k_linear_f16 — the rest)"""
    N, K = w.shape
    y = out if out is not None else torch.empty(x2.shape[0], N, dtype=torch.float16, device=x2.device)
    if residual is not None and (residual.stride(0) != y.stride(0) or residual.dtype != torch.float16):
        raise RuntimeError("_linear_f16: the residual must share the output's row pitch")
    rf = (0 if relu else N) if relu_from is None else relu_from
    if RS_GEMM and L.lib().devo_upd_rs_supported(N, K) and y.stride(0) % 8 == 0 and y.data_ptr() % 16 == 0 and y.stride(1) == 1:
        L.check(L.lib().devo_upd_rs_linear_f16(L.ptr(x2), x2.stride(0), L.ptr(_rs_image(w.detach())), L.ptr(b), L.ptr(residual), L.ptr(y), y.stride(0),
                                               x2.shape[0], N, K, rf, L.stream()), "update.rs_linear_f16")
        return y
    L.check(L.lib().devo_upd_linear_f16(L.ptr(x2), x2.stride(0), L.ptr(_split_weight(w.detach(), False)), L.ptr(b), L.ptr(residual), L.ptr(y),
                                        y.stride(0), x2.shape[0], N, K, rf, L.stream()), "update.linear_f16")
    return y


MLP2_F16 = __import__("os").environ.get("DEVO_UPD_MLP2", "1") != "0"            # 0: Linear - ReLU - Linear chains as two launches


def _mlp2_image(w):
    """The B-operand image of csrc/mlp2.hip for a [384, K] fp16 weight, cached per version like _split_weight's images."""
    key = (w.data_ptr(), w._version, tuple(w.shape), tuple(w.stride()), "mlp2")
    hit = _wsplit_cache.pop(key, None)
    if hit is not None:
        _wsplit_cache[key] = hit
        return hit[1]
    for k in [k for k in _wsplit_cache if k[0] == key[0] and k[4] == "mlp2"]:
        del _wsplit_cache[k]
    while len(_wsplit_cache) >= WSPLIT_CACHE_ENTRIES:
        del _wsplit_cache[next(iter(_wsplit_cache))]
    N, K = w.shape
    img = torch.empty(int(L.lib().devo_upd_mlp2_weight_bytes(K)) // 2, dtype=torch.float16, device=w.device)
    L.check(L.lib().devo_upd_mlp2_pack_weight(L.ptr(w), w.stride(0), w.stride(1), K, L.ptr(img), L.stream()), "update.mlp2_pack_weight")
    _wsplit_cache[key] = (w, img)
    return img


def _mlp2_ok(x2, l1, l2):
    return (MLP2_F16 and x2.is_cuda and x2.dtype == torch.float16 and l1.weight.dtype == torch.float16 and l2.weight.dtype == torch.float16
            and l1.weight.shape[0] == 384 and tuple(l2.weight.shape) == (384, 384) and x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) % 2 == 0
            and x2.data_ptr() % 4 == 0 and x2.shape[1] == l1.weight.shape[1] and x2.shape[0] >= 1024
            and ((x2.shape[0] - 1) * x2.stride(0) + x2.shape[1]) * 2 < (1 << 31))


def _mlp2_f16(x2, l1, l2, residual=None, gather=None, fg=None):
    """l2(relu(l1(x2[gather]))) [+ residual] as one launch (384 inputs: csrc/gemm_rs.hip, the rows resident in LDS; else csrc/mlp2.hip);
    gather i64 [rows] with negative entries = zero rows"""
    rows = x2.shape[0] if gather is None else gather.numel()
    y = torch.empty(rows, 384, dtype=torch.float16, device=x2.device)
    if residual is not None and (residual.stride(0) != 384 or residual.stride(1) != 1 or residual.dtype != torch.float16 or residual.data_ptr() % 16):
        raise RuntimeError("_mlp2_f16: the residual must be a contiguous fp16 [rows, 384] tensor")
    if RS_CHAINS and RS_GEMM and l1.weight.shape[1] == 384 and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0 and l1.bias.data_ptr() % 8 == 0 \
            and l2.bias.data_ptr() % 8 == 0:
        # fg = (W [768, 384], b [768]): the f | g layer of the SoftAgg that follows, on the result rows while they are in LDS -> (y, fg [rows, 768])
        fg_out = torch.empty(rows, 768, dtype=torch.float16, device=x2.device) if fg is not None else None
        L.check(L.lib().devo_upd_rs_mlp2_fg_f16(L.ptr(x2), x2.stride(0), x2.shape[0], L.ptr(gather), L.ptr(_rs_image(l1.weight.detach())), L.ptr(l1.bias),
                                                L.ptr(_rs_image(l2.weight.detach())), L.ptr(l2.bias), L.ptr(residual), L.ptr(y), rows,
                                                L.ptr(_rs_image(fg[0])) if fg is not None else None, L.ptr(fg[1]) if fg is not None else None, L.ptr(fg_out),
                                                L.stream()), "update.rs_mlp2_f16")
        return y if fg is None else (y, fg_out)
    if fg is not None:
        raise RuntimeError("_mlp2_f16: the f | g tail needs the row-resident kernel")
    L.check(L.lib().devo_upd_mlp2_f16(L.ptr(x2), x2.stride(0), x2.shape[0], L.ptr(gather), L.ptr(_mlp2_image(l1.weight.detach())), L.ptr(l1.bias),
                                      l1.weight.shape[1], L.ptr(_mlp2_image(l2.weight.detach())), L.ptr(l2.bias), L.ptr(residual), L.ptr(y), 384, rows,
                                      L.stream()), "update.mlp2_f16")
    return y


MIXED_STATE = __import__("os").environ.get("DEVO_UPD_MIXED_STATE", "1") != "0"      # 0: the autocast call converts the fp32 state with torch in front of / behind the fp16 operator
AUTOCAST_F16 = __import__("os").environ.get("DEVO_UPD_AUTOCAST_F16", "1") != "0"    # 0: an fp32 operator called under autocast keeps its fp32 kernels
RS_SPLIT = __import__("os").environ.get("DEVO_UPD_RS_SPLIT", "1") != "0"         # 0: every fp32 Linear layer on csrc/linear.hip's kernel


def _rs_split_image(w, transposed):
    """The split B-operand image of csrc/gemm_rs.hip for `x @ w.T` (transposed=False) or `g @ w` (True), cached per version like _split_weight's."""
    key = (w.data_ptr(), w._version, tuple(w.shape), tuple(w.stride()), ("rs32", transposed))
    hit = _wsplit_cache.pop(key, None)
    if hit is not None:
        _wsplit_cache[key] = hit
        return hit[1]
    for k in [k for k in _wsplit_cache if k[0] == key[0] and k[4] == key[4]]:
        del _wsplit_cache[k]
    while len(_wsplit_cache) >= WSPLIT_CACHE_ENTRIES:
        del _wsplit_cache[next(iter(_wsplit_cache))]
    out_f, in_f = w.shape
    N, K = (in_f, out_f) if transposed else (out_f, in_f)
    s_n, s_k = (w.stride(1), w.stride(0)) if transposed else (w.stride(0), w.stride(1))
    img = torch.empty(int(L.lib().devo_upd_rs_split_weight_bytes(N, K)) // 4, dtype=torch.float32, device=w.device)
    L.check(L.lib().devo_upd_rs_split_weight(L.ptr(w), s_n, s_k, N, K, L.ptr(img), L.stream()), "update.rs_split_weight")
    _wsplit_cache[key] = (w, img)
    return img


def _linear_split(x2, w, b, transposed=False, relu=False, relu_from=None, residual=None, out=None, gate=None, rs=False):
    """x2 [rows, K] fp32 -> act(x2 @ w.T + b) [+ residual] (or x2 @ w: transposed) on the fp16 matrix cores, fp32 in and out
    (csrc/linear.hip).  relu_from: the ReLU from this output column on (a gate | res pair in one launch); residual / out: [rows, N]
    fp32 with one row pitch (out may be the residual: x.add_(linear(t)) in one launch); gate [rows, N]: a ReLU's output — the result is
    zeroed where it clipped (the ReLU's adjoint in the epilogue of the dX product that produces its incoming gradient).  rs: the inference
    operator's calls — 384-wide layers over >= 8192 aligned rows take csrc/gemm_rs.hip's row-resident kernel."""
    N = w.shape[1] if transposed else w.shape[0]
    K = w.shape[0] if transposed else w.shape[1]
    y = out if out is not None else torch.empty(x2.shape[0], N, dtype=torch.float32, device=x2.device)
    for t, what in ((residual, "residual"), (gate, "gate")):
        if t is not None and (t.stride(0) != y.stride(0) or t.stride(1) != 1 or t.dtype != torch.float32):
            raise RuntimeError(f"_linear_split: the {what} must share the output's row pitch")
    rf = (0 if relu else N) if relu_from is None else relu_from
    # (measured: 27.4 against 29.9 us per 18 000 x 384 x 384 layer, the inference operator 0.716 against 0.745 ms; nothing for the training step,
    #  whose layers run under autograd with their other shapes: those stay on linear.hip's kernel)
    if (RS_SPLIT and N == 384 and x2.shape[0] >= 8192 and rs and L.lib().devo_upd_rs_split_supported(N, K) and x2.stride(0) % 4 == 0 and y.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0
            and y.data_ptr() % 16 == 0 and (residual is None or residual.data_ptr() % 16 == 0) and (gate is None or gate.data_ptr() % 16 == 0)):
        L.check(L.lib().devo_upd_rs_linear_split(L.ptr(x2), x2.stride(0), L.ptr(_rs_split_image(w.detach(), transposed)), L.ptr(b), L.ptr(residual),
                                                 L.ptr(gate), L.ptr(y), y.stride(0), x2.shape[0], N, K, rf, L.stream()), "update.rs_linear_split")
        return y
    L.check(L.lib().devo_upd_linear_split(L.ptr(x2), x2.stride(0), L.ptr(_split_weight(w.detach(), transposed)), L.ptr(b), L.ptr(residual),
                                          L.ptr(gate), L.ptr(y), y.stride(0), x2.shape[0], N, K, rf, L.stream()), "update.linear_split")
    return y


_dw_ws = {}                 # (device, stream) -> workspace of the weight-gradient kernel (grown on demand, reused: ONE stream serialises its users)
SPLIT_DW = __import__("os").environ.get("DEVO_UPD_SPLIT_DW", "1") != "0"        # 0: the library's products for dW / db


def _dw_ok(g2, x2):
    return (SPLIT_GEMM and SPLIT_DW and g2.is_cuda and g2.dtype == x2.dtype == torch.float32 and g2.shape[1] >= 96 and x2.shape[1] >= 96
            and x2.stride(1) == 1 and x2.stride(0) >= x2.shape[1] and g2.shape[0] >= 2048
            and g2.shape[0] * max(g2.shape[1], x2.stride(0)) * 4 < (1 << 31))


def _dw_split(g2, x2, with_bias):
    """(dW [No, Ni], db [No] or None) = (g2^T x2, column sums of g2): fp32 in and out, exact hi + lo splits on the fp16 matrix cores"""
    R, No, Ni = g2.shape[0], g2.shape[1], x2.shape[1]
    need = L.lib().devo_upd_dw_workspace_bytes(R, No, Ni)
    wkey = (g2.device, L.stream().value)
    ws = _dw_ws.get(wkey)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=g2.device)
        _dw_ws[wkey] = ws
    dW = torch.empty(No, Ni, dtype=torch.float32, device=g2.device)
    db = torch.empty(No, dtype=torch.float32, device=g2.device) if with_bias else None
    L.check(L.lib().devo_upd_dw_split(L.ptr(g2), g2.stride(0), L.ptr(x2), x2.stride(0), R, No, Ni, L.ptr(ws), L.ptr(dW), Ni, L.ptr(db), L.stream()),
            "update.dw_split")
    return dW, db


class _LinearFn(torch.autograd.Function):
    """y = act(x Wᵀ + b) [+ residual] over many rows (the Update operator sees one row per edge: 18 000 at BASELINE configuration 3), all
    three products of its training step on the fp16 matrix cores with exact hi + lo splits of every fp32 value (2^-22 relative per factor,
    fp32 accumulation, power-of-two scales against fp16's range): y and dX = dY W through csrc/linear.hip (any widths >= 96 outputs / 32
    inputs, _split_ok; 29.9 us against the library's 63 us for 18 000 x 384 x 384), dW = dYᵀ X and db through csrc/linear_dw.hip (any widths
    >= 96, _dw_ok: the corr MLP's 882 inputs included; 50 against 98 us).  Layers below those widths (the 2-wide heads) use the library: for
    dW as a batched product over 16 row chunks + a sum (74 us where the direct product runs on a handful of tiles, 148 us;
    tools/ubench_dw_gemm.py)."""
    CHUNKS = 16

    @staticmethod
    def forward(ctx, x, w, b, relu=False, residual=None):
        """relu: max(., 0) behind the layer (the Sequential(Linear, ReLU, ...) pairs of enet.py / blocks.py: the GEMM's epilogue);
        residual [rows, out]: added to the result (`net + c(t)`, enet.py:88-91: the epilogue again)"""
        ctx.has_bias, ctx.relu, ctx.has_res = b is not None, bool(relu), residual is not None
        if x.dtype == w.dtype == torch.float32 and (b is None or b.dtype == torch.float32) and _split_ok(x, w.shape[0], w.shape[1]) \
                and not torch.is_autocast_enabled() and (residual is None or residual.dtype == torch.float32):
            y = _linear_split(x, w, b.contiguous() if b is not None else None, relu=relu,
                              residual=residual.contiguous() if residual is not None else None)
        else:
            y = torch.nn.functional.linear(x, w, b)
            if relu:
                y = torch.relu_(y)
            if residual is not None:
                y = y + residual.to(y.dtype)
        if relu and residual is not None:
            raise RuntimeError("_LinearFn: relu and residual together are not needed by the operator")
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        gx = gw = gb = None
        gres = g if ctx.has_res and ctx.needs_input_grad[4] else None
        if ctx.relu:
            g = torch.ops.aten.threshold_backward(g.contiguous(), y, 0)       # no gradient where the ReLU clipped
        g2 = g.reshape(-1, g.shape[-1])
        # under torch.autocast (devo.py:311 runs the update operator under it) the forward's output — and so g — is fp16 while x and
        # the parameters are fp32: multiply in g's dtype like the forward did, return every gradient in its input's dtype
        with torch.autocast(device_type="cuda", enabled=False):
            if ctx.needs_input_grad[0]:
                if g2.dtype == w.dtype == x.dtype == torch.float32 and _split_ok(g2.contiguous(), w.shape[1], w.shape[0]):
                    gx = _linear_split(g2.contiguous(), w, None, transposed=True).reshape(x.shape)
                else:
                    gx = (g2 @ w.to(g2.dtype)).reshape(x.shape).to(x.dtype)
            need_w, need_b = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
            x2 = x.reshape(-1, x.shape[-1])
            if need_w and _dw_ok(g2, x2):                                   # dW and db in one pass on the fp16 matrix cores (csrc/linear_dw.hip)
                gw, gb = _dw_split(g2 if g2.is_contiguous() else g2.contiguous(), x2, need_b)
            else:
                if need_w:
                    x2 = x2.to(g2.dtype)
                    rows, S = x2.shape[0], _LinearFn.CHUNKS
                    if rows % S == 0:
                        gw = torch.bmm(g2.reshape(S, rows // S, -1).transpose(1, 2), x2.reshape(S, rows // S, -1)).sum(0)
                    else:
                        gw = g2.t() @ x2
                    gw = gw.to(w.dtype)
                if need_b:
                    gb = g2.sum(0).to(w.dtype)
        return gx, gw, gb, None, gres


def _linear(x2, w, b):
    """x2 [rows, in] -> x2 wT + b through _LinearFn when that pays (see Linear.forward)"""
    if torch.is_grad_enabled() and x2.is_cuda and x2.shape[0] >= 4096 and (x2.requires_grad or w.requires_grad):
        return _LinearFn.apply(x2, w, b)
    return torch.nn.functional.linear(x2, w, b)


class _Mlp2Fn(torch.autograd.Function):
    """Linear -> ReLU -> Linear [+ residual] (the c1 / c2 / res / corr pairs of enet.py:41-66, blocks.py:15-29) as four split-precision
    products per direction with everything between them in their epilogues: forward relu(x W1^T + b1), then h W2^T + b2 + residual;
    backward dH = (dY W2) masked by h > 0 (the ReLU's adjoint in the dX product's epilogue: no threshold_backward pass), dW2 / db2 from
    (dY, h), dX = dH W1, dW1 / db1 from (dH, x)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual):
        h = _linear_split(x, w1, b1.contiguous() if b1 is not None else None, relu=True)
        y = _linear_split(h, w2, b2.contiguous() if b2 is not None else None, residual=residual.contiguous() if residual is not None else None)
        ctx.save_for_backward(x, h, w1, w2)
        ctx.bias = (b1 is not None, b2 is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, h, w1, w2 = ctx.saved_tensors
        need = ctx.needs_input_grad
        g = g.contiguous()
        gx = gw1 = gb1 = gw2 = gb2 = None
        gh = _linear_split(g, w2, None, transposed=True, gate=h)
        if need[3] or (ctx.bias[1] and need[4]):
            gw2, gb2 = _dw_split(g, h, ctx.bias[1] and need[4])
        if need[0]:
            gx = _linear_split(gh, w1, None, transposed=True)
        if need[1] or (ctx.bias[0] and need[2]):
            gw1, gb1 = _dw_split(gh, x, ctx.bias[0] and need[2])
        return gx, gw1, gb1, gw2, gb2, (g if need[5] else None)


def _mlp2_fused_ok(seq, x2, residual):
    w1, w2 = seq[0].weight, seq[2].weight
    if not (w1.dtype == w2.dtype == torch.float32 and x2.is_contiguous() and (residual is None or residual.dtype == torch.float32)):
        return False
    rows = x2.shape[0]
    ok_fwd = _split_ok(x2, w1.shape[0], w1.shape[1]) and w2.shape[1] >= 32 and w2.shape[0] >= 96 and w1.shape[0] >= 96
    ok_dw = SPLIT_DW and min(w1.shape[0], w1.shape[1], w2.shape[0], w2.shape[1]) >= 96 and rows >= 2048
    return ok_fwd and ok_dw


FUSE_EPILOGUE = __import__("os").environ.get("DEVO_UPD_FUSE_EPILOGUE", "1") != "0"     # 0: ReLU / residual sums as ATen kernels in the training path


def _mlp2(seq, x, residual=None):
    """Sequential(Linear, ReLU, Linear)(x) [+ residual] for the autograd path: the ReLU and the residual sum in the GEMMs' epilogues when
    the rows are the operator's (>= 4096 fp32 rows on the GPU), the modules as they are otherwise"""
    rows = x.numel() // x.shape[-1]
    if (FUSE_EPILOGUE and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and rows >= 4096 and not torch.is_autocast_enabled()
            and (x.requires_grad or seq[0].weight.requires_grad)):
        x2, r2 = x.reshape(rows, -1), (residual.reshape(rows, -1) if residual is not None else None)
        if _mlp2_fused_ok(seq, x2, r2):
            y = _Mlp2Fn.apply(x2, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias, r2)
        else:
            h = _LinearFn.apply(x2, seq[0].weight, seq[0].bias, True, None)
            y = _LinearFn.apply(h, seq[2].weight, seq[2].bias, False, r2)
        return y.view(*x.shape[:-1], y.shape[-1])
    y = seq[2](seq[1](seq[0](x)))
    return y if residual is None else residual + y


# A Linear / LayerNorm of this file that gets a NEW Parameter object (`m.weight = nn.Parameter(...)`) bumps the epoch cell of the Update
# operator it belongs to (Update.__init__ hands every layer its cell): the operator's plans hold raw pointers of the parameters they were
# made from and key themselves on (that epoch, every parameter's storage address and version counter) — an edit in place moves a version,
# `.data = ...` / .to() / .half() move an address, a replaced Parameter moves the epoch.  (A cell per operator, not one per process: building
# another operator — the fp16 shadow of an autocast call is one — must not age the plans of this one.)
def _bump_epoch(module, name):
    if name in ("weight", "bias"):
        cell = module.__dict__.get("_epoch_cell")
        if cell is not None:
            cell[0] += 1


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) whose backward over >= 4096 rows splits the weight-gradient product."""

    def __setattr__(self, name, value):
        _bump_epoch(self, name)
        super().__setattr__(name, value)

    def forward(self, x):
        if torch.is_grad_enabled() and x.is_cuda and x.numel() // x.shape[-1] >= 4096 and (x.requires_grad or self.weight.requires_grad):
            y = _LinearFn.apply(x.reshape(-1, x.shape[-1]), self.weight, self.bias)      # (2-D inside: a Function's output must not be a view for the in-place ReLUs)
            return y.view(*x.shape[:-1], y.shape[-1])
        return super().forward(x)


class _LayerNormFn(torch.autograd.Function):
    """y = LN(x + a + b) [ReLU'd] over 384-wide fp32 rows with the HIP kernels in both directions (devo_upd_layernorm /
    devo_upd_layernorm_backward): one pass per direction, the sums in front of the norm and the ReLU behind it fused in, the
    column sums for gamma / beta folded inside the workgroups.  ATen: two adds, the norm, a clamp; grad_input + two kernels for the
    parameter gradients + the clamp's and the adds' adjoints."""

    @staticmethod
    def forward(ctx, x, a, b, weight, bias, eps, relu):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        a2 = a.reshape(-1, x.shape[-1]).contiguous() if a is not None else None
        b2 = b.reshape(-1, x.shape[-1]).contiguous() if b is not None else None
        out = torch.empty_like(x2)
        L.check(L.lib().devo_upd_layernorm(L.ptr(x2), L.ptr(a2), L.ptr(b2), None, None, None, 0, None, L.ptr(weight), L.ptr(bias), L.ptr(out),
                                           x2.shape[0], x2.shape[1], float(eps), int(relu), L.dtype_code(x2), L.stream()), "update.layernorm")
        ctx.save_for_backward(x2, a2, b2, weight, bias)
        ctx.eps, ctx.relu, ctx.shape = float(eps), bool(relu), x.shape
        return out.view(x.shape) if out.shape != x.shape else out      # (not a view where the shape allows: an in-place ReLU may follow)

    @staticmethod
    def backward(ctx, g):
        x2, a2, b2, weight, bias = ctx.saved_tensors
        g2 = g.reshape(x2.shape).to(torch.float32).contiguous()
        dx = torch.empty_like(x2)
        dwb = torch.zeros(2, x2.shape[1], dtype=torch.float32, device=x2.device)
        # torch.use_deterministic_algorithms(True): the workgroups' column sums through a scratch and a second kernel, in a fixed order
        # (bit-reproducible gamma / beta gradients, like ATen's two-stage reduction); else float atomics (one launch less)
        part = torch.empty(512 * 2 * x2.shape[1], dtype=torch.float32, device=x2.device) if torch.are_deterministic_algorithms_enabled() else None
        L.check(L.lib().devo_upd_layernorm_backward(L.ptr(x2), L.ptr(a2), L.ptr(b2), L.ptr(weight), L.ptr(bias), L.ptr(g2), L.ptr(dx), L.ptr(dwb[0]),
                                                    L.ptr(dwb[1]), x2.shape[0], x2.shape[1], ctx.eps, int(ctx.relu), L.ptr(part), L.stream()),
                "update.layernorm_backward")
        dxv = dx.view(ctx.shape)
        return (dxv if ctx.needs_input_grad[0] else None, dxv if a2 is not None and ctx.needs_input_grad[1] else None,
                dxv if b2 is not None and ctx.needs_input_grad[2] else None, dwb[0] if ctx.needs_input_grad[3] else None,
                dwb[1] if ctx.needs_input_grad[4] else None, None, None)


HIP_LAYERNORM = __import__("os").environ.get("DEVO_UPD_HIP_LAYERNORM", "1") != "0"      # 0: ATen's LayerNorm in the training path


def _ln_train(mod, x, a=None, b=None, relu=False):
    """mod(x + a + b) [ReLU'd] for the autograd path: the HIP pair when the rows are the operator's (fp32, 384 wide, on the GPU)"""
    if (HIP_LAYERNORM and x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 384 and mod.weight.dtype == torch.float32
            and not torch.is_autocast_enabled() and all(t is None or (t.dtype == torch.float32 and t.shape == x.shape) for t in (a, b))):
        return _LayerNormFn.apply(x, a, b, mod.weight, mod.bias, mod.eps, relu)
    if a is not None:
        x = x + a
    if b is not None:
        x = x + b
    y = mod(x)
    return torch.relu(y) if relu else y


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters, same state-dict keys) whose training path runs the HIP forward / backward pair"""

    def __setattr__(self, name, value):
        _bump_epoch(self, name)
        super().__setattr__(name, value)

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            return _ln_train(_PlainLN(self), x)
        return super().forward(x)


class _PlainLN:
    """what _ln_train needs of a LayerNorm module (weight, bias, eps, the ATen call) without re-entering LayerNorm.forward"""

    def __init__(self, mod):
        self.weight, self.bias, self.eps, self._shape = mod.weight, mod.bias, mod.eps, mod.normalized_shape

    def __call__(self, x):
        return F.layer_norm(x, self._shape, self.weight, self.bias, self.eps)


class GradientClip(nn.Module):                       # blocks.py:84-89
    def forward(self, x):
        return _GradClip.apply(x)


class _GatedResidualFn(torch.autograd.Function):
    """x + sigmoid(gate) * res, one HIP kernel per direction (devo_upd_gated_residual / _backward) instead of sigmoid, mul, add and
    their five autograd kernels."""

    @staticmethod
    def forward(ctx, x, gate, res):
        if not (x.dtype == gate.dtype == res.dtype):
            raise RuntimeError("update.gated_residual: x, gate and res must share one dtype (the kernel reads all three with it)")
        x, gate, res = x.contiguous(), gate.contiguous(), res.contiguous()
        out = torch.empty_like(x)
        rows, dim = x.numel() // x.shape[-1], x.shape[-1]
        L.check(L.lib().devo_upd_gated_residual(L.ptr(x), L.ptr(gate), dim, L.ptr(res), L.ptr(out), rows, dim, L.dtype_code(x), L.stream()),
                "update.gated_residual")
        ctx.save_for_backward(gate, res)
        return out

    @staticmethod
    def backward(ctx, g):
        gate, res = ctx.saved_tensors
        g = g.to(gate.dtype).contiguous()
        dgate, dres = torch.empty_like(gate), torch.empty_like(res)
        rows, dim = g.numel() // g.shape[-1], g.shape[-1]
        L.check(L.lib().devo_upd_gated_residual_backward(L.ptr(gate), dim, L.ptr(res), L.ptr(g), L.ptr(dgate), L.ptr(dres), rows, dim,
                                                         L.dtype_code(g), L.stream()), "update.gated_residual_backward")
        return g, dgate, dres


class _MaskedGatherFn(torch.autograd.Function):
    """out[e] = src[idx[e]] where idx[e] >= 0, else 0 (`mask * net[:, ix]`, enet.py:87-91).  idx / inv are the previous- / next-edge
    maps of `neighbors` (mutually inverse partial permutations), so the adjoint of the gather by idx is the gather by inv."""

    @staticmethod
    def forward(ctx, src, idx, inv):
        ctx.save_for_backward(inv)
        return _MaskedGatherFn._gather(src, idx)

    @staticmethod
    def _gather(src, idx):
        src = src.contiguous()
        out = torch.empty_like(src)
        L.check(L.lib().devo_upd_masked_gather(L.ptr(src), L.ptr(idx), L.ptr(out), src.shape[0], src.shape[1], L.dtype_code(src), L.stream()),
                "update.masked_gather")
        return out

    @staticmethod
    def backward(ctx, g):
        inv, = ctx.saved_tensors
        return _MaskedGatherFn._gather(g, inv), None, None


def _hip_ok(x):
    return x.is_cuda and x.dtype in (torch.float32, torch.float16)


class GatedResidual(nn.Module):                      # blocks.py:15-29
    def __init__(self, dim):
        super().__init__()
        self.gate = nn.Sequential(Linear(dim, dim), nn.Sigmoid())
        self.res = nn.Sequential(Linear(dim, dim), nn.ReLU(inplace=True), Linear(dim, dim))

    def forward(self, x):
        if _hip_ok(x):
            gate, res = self.gate[0](x), _mlp2(self.res, x)
            if gate.dtype == x.dtype and res.dtype == x.dtype:   # (under autocast the Linear outputs are fp16 next to an fp32 x: torch expression)
                return _GatedResidualFn.apply(x, gate, res)
            return x + torch.sigmoid(gate) * res
        return x + self.gate(x) * self.res(x)


class SoftAgg(nn.Module):                            # blocks.py:31-48 (expand=True)
    def __init__(self, dim=512):
        super().__init__()
        self.f = Linear(dim, dim)
        self.g = Linear(dim, dim)
        self.h = Linear(dim, dim)

    def forward(self, x, ix, groups=None):           # torch composition (autograd path)
        """`groups` = (inverse, count) of torch.unique(ix): the caller may cache them per graph (the sort and the host read of the
        count are per call otherwise)"""
        if isinstance(groups, _Groups):
            # fp32 / fp16 on the GPU: f | g in ONE GEMM, the segment softmax-sum and its adjoint as HIP kernels, h, expand
            B, E, C = x.shape
            fg = _linear(x.reshape(E, C), torch.cat([self.f.weight, self.g.weight], 0), torch.cat([self.f.bias, self.g.bias], 0))
            y = _SoftAggFn.apply(fg, groups)
            return torch.index_select(self.h(y), 0, groups.group_of_long()).view(B, E, C)
        if groups is None:
            _, jx = torch.unique(ix, return_inverse=True)
            n = int(jx.max()) + 1
        else:
            jx, n = groups
        B, E, C = x.shape
        idx = jx.view(1, E, 1).expand(B, E, C)
        gx, fx = self.g(x), self.f(x)
        mx = torch.full((B, n, C), float("-inf"), dtype=x.dtype, device=x.device).scatter_reduce(1, idx, gx, "amax", include_self=True)
        ex = (gx - mx.gather(1, idx)).exp()
        w = ex / torch.zeros(B, n, C, dtype=x.dtype, device=x.device).scatter_add(1, idx, ex).gather(1, idx)
        y = torch.zeros(B, n, C, dtype=x.dtype, device=x.device).scatter_add(1, idx, fx * w)
        return torch.index_select(self.h(y), 1, jx)        # (index_select: its backward is an atomic index_add; `[:, jx]` sorts)


class _SoftAggFn(torch.autograd.Function):
    """y[s] = sum over the edges e of group s of f_e * softmax_s(g)_e (blocks.py:42-43) with the HIP kernels in both directions:
    devo_upd_softagg / devo_upd_softagg_backward.  fg [E, 2 dim] = f | g (one GEMM); G: the group tables (_Groups)."""

    @staticmethod
    def forward(ctx, fg, G):
        E, dim = fg.shape[0], fg.shape[1] // 2
        fg = fg.contiguous()
        y = torch.empty(G.n_seg, dim, dtype=fg.dtype, device=fg.device)
        L.check(L.lib().devo_upd_softagg_hint(L.ptr(fg), L.ptr(fg[:, dim:]), 2 * dim, L.ptr(G.perm), L.ptr(G.seg_start), L.ptr(G.n_seg_dev),
                                              L.ptr(y), None, E, dim, L.dtype_code(fg), int(E // max(int(G.n_seg), 1)), L.stream()), "update.softagg")
        ctx.save_for_backward(fg)
        ctx.G = G
        return y

    @staticmethod
    def backward(ctx, dy):
        fg, = ctx.saved_tensors
        G = ctx.G
        E, dim = fg.shape[0], fg.shape[1] // 2
        dy = dy.to(fg.dtype).contiguous()
        dfg = torch.empty_like(fg)                              # (every edge belongs to exactly one group: fully written)
        L.check(L.lib().devo_upd_softagg_backward(L.ptr(fg), L.ptr(fg[:, dim:]), 2 * dim, L.ptr(G.perm), L.ptr(G.seg_start), L.ptr(G.n_seg_dev),
                                                  L.ptr(dy), L.ptr(dfg), L.ptr(dfg[:, dim:]), 2 * dim, E, dim, L.dtype_code(fg), L.stream()),
                "update.softagg_backward")
        return dfg, None


_PINNED = {"pool": [], "next": 0}


def _pinned_word():
    """A pinned int32 word + an event from a small ring (hipHostMalloc per graph costs ~100 us of host time).  A slot is reused after 32 other
    graphs; its previous owner has read it by then or never will."""
    P = _PINNED
    if len(P["pool"]) < 32:
        P["pool"].append((torch.empty(1, dtype=torch.int32).pin_memory(), torch.cuda.Event()))
        return P["pool"][-1]
    P["next"] = (P["next"] + 1) % 32
    return P["pool"][P["next"]]


_GRAPH_TABLES = os.environ.get("DEVO_UPD_GRAPH_TABLES", "1") != "0"     # (0: the separate calls of rounds 1-5, for A/B runs)


class _Groups:
    """Edges grouped by an integer key, through the BA's index kernels: perm / seg_start / n_seg (+ group_of scratch)."""

    def __init__(self, key, bound=None):
        """round 6: no host synchronisation here.  `bound` only sizes the workspace (any value above the largest key will do: the index kernels
        work on the range of keys the list holds); the number of groups stays on the device (`n_seg_dev`, what the kernels read) and is copied to
        a pinned host word behind the preparation — `n_seg` waits for THAT copy when the first caller asks for the integer (an allocation size, a
        GEMM's row count), by which time the stream has long moved on: DEVO's steady state builds these tables for a new graph every frame, and
        three stalls per frame (the bounds, two group counts) cost more than the tables."""
        E = key.numel()
        if bound is None:
            bound = 1 << 20
        ws = cuda_ba.workspace(E, bound, 0, key.device)
        cuda_ba.prepare(key, bound, 0, ws)
        self.n_seg_dev, _, self.seg_start, self.perm = cuda_ba.table_views(ws, E, bound, 0)     # (views: `ws` is this object's, nobody prepares it again)
        self._finish(E)

    @classmethod
    def from_tables(cls, tables, E):
        """tables = (n_seg i32 [1], seg_start, perm) of cuda_ba.graph_tables: device views of a workspace that call allocated"""
        self = cls.__new__(cls)
        self.n_seg_dev, self.seg_start, self.perm = tables
        self._finish(E)
        return self

    def _finish(self, E):
        key = self.perm
        self._n_host, self._n_event = _pinned_word()
        self._n_host.copy_(self.n_seg_dev, non_blocking=True)
        self._n_event.record()
        self._n_seg = None
        self.n_edges = E
        self.group_of = torch.empty(E, dtype=torch.int32, device=key.device)
        self._group_of_long = None

    @property
    def n_seg(self):
        """The number of groups as a Python int: waits for the copy behind the preparation (the training path's allocations need it)."""
        if self._n_seg is None:
            self._n_event.synchronize()
            self._n_seg = int(self._n_host[0])
        return self._n_seg

    def rows(self):
        """Rows to allocate / launch for per-group tensors WITHOUT waiting: the number of groups if its copy has arrived, else the number of
        edges (an upper bound: the kernels read the true count from `n_seg_dev`, the GEMM behind them multiplies a few thousand idle rows —
        6 us — where waiting for the count drains the whole eager pipeline, ~0.5 ms per frame in DEVO's steady state)."""
        if self._n_seg is None and self._n_event.query():
            self._n_seg = int(self._n_host[0])
        return self._n_seg if self._n_seg is not None else self.n_edges


    def group_of_long(self):
        """edge -> group index as int64 (for index_select in the training path), from the tables themselves"""
        if self._group_of_long is None:
            cnt = (self.seg_start[1:self.n_seg + 1] - self.seg_start[:self.n_seg]).long()
            g = torch.empty(self.perm.numel(), dtype=torch.int64, device=self.perm.device)
            g[self.perm.long()] = torch.repeat_interleave(torch.arange(self.n_seg, device=self.perm.device), cnt)
            self._group_of_long = g
        return self._group_of_long


def _ln(x, mod, add1=None, add2=None, expand=None, gated=None, relu=False):
    """LayerNorm(x + add1 + add2 + hy[group_of] + sigmoid(gate) * res), optionally ReLU'd — one fused kernel.
    expand = (hy, group_of); gated = (gate, res)."""
    out = torch.empty_like(x)
    hy, grp = expand if expand is not None else (None, None)
    gate, res = gated if gated is not None else (None, None)
    rc = L.lib().devo_upd_layernorm(L.ptr(x), L.ptr(add1), L.ptr(add2), L.ptr(hy), L.ptr(grp), L.ptr(gate),
                                    gate.stride(0) if gate is not None else 0, L.ptr(res), L.ptr(mod.weight), L.ptr(mod.bias), L.ptr(out),
                                    x.shape[0], x.shape[1], float(mod.eps), int(relu), L.dtype_code(x), L.stream())
    L.check(rc, "update.layernorm")
    return out


class Update(nn.Module):
    def __init__(self, p, dim=DIM):
        super().__init__()
        self.dim = dim
        self.c1 = nn.Sequential(Linear(dim, dim), nn.ReLU(inplace=True), Linear(dim, dim))
        self.c2 = nn.Sequential(Linear(dim, dim), nn.ReLU(inplace=True), Linear(dim, dim))
        self.norm = LayerNorm(dim, eps=1e-3)
        self.agg_kk = SoftAgg(dim)
        self.agg_ij = SoftAgg(dim)
        self.gru = nn.Sequential(LayerNorm(dim, eps=1e-3), GatedResidual(dim), LayerNorm(dim, eps=1e-3), GatedResidual(dim))
        self.corr = nn.Sequential(Linear(2 * 49 * p * p, dim), nn.ReLU(inplace=True), Linear(dim, dim),
                                  LayerNorm(dim, eps=1e-3), nn.ReLU(inplace=True), Linear(dim, dim))
        self.d = nn.Sequential(nn.ReLU(inplace=False), Linear(dim, 2), GradientClip())
        self.w = nn.Sequential(nn.ReLU(inplace=False), Linear(dim, 2), GradientClip(), nn.Sigmoid())
        self._graph_key, self._graph, self._graph_refs, self._wcat = None, None, None, {}
        self.__dict__["_epoch"] = [0]                                     # (see _bump_epoch)
        for m in self.modules():
            if isinstance(m, (Linear, LayerNorm)):
                m.__dict__["_epoch_cell"] = self.__dict__["_epoch"]

    # ------------------------------------------------------------------------------------------ torch / autograd path
    def forward_torch(self, net, inp, corr, ii, jj, kk):
        """enet.py:80-99 as a torch composition over GPU tensors (differentiable)."""
        # corr MLP (enet.py:59-66): its LayerNorm + ReLU as one kernel per direction; net = norm(net + inp + corr) with the sums inside
        c = _mlp2(self.corr, corr)                                         # (corr[0], ReLU, corr[2])
        c = self.corr[5](_ln_train(_PlainLN(self.corr[3]), c, relu=True))
        net = _ln_train(_PlainLN(self.norm), net, inp, c)
        if net.is_cuda and net.dtype in (torch.float32, torch.float16) and net.shape[0] == 1:
            ix, jx, gk, gp = self._tables(ii, jj, kk)               # neighbours + HIP group tables, cached per graph
        else:
            (ix, jx), gk, gp = self._torch_groups(ii, jj, kk)       # ... + torch.unique group maps (any dtype / batch)
        mask_ix = (ix >= 0).to(net.dtype).reshape(1, -1, 1)
        mask_jx = (jx >= 0).to(net.dtype).reshape(1, -1, 1)
        # gathers with index_select (backward = atomic index_add; advanced indexing's backward sorts 18 000 indices: 0.43 ms each)
        if _hip_ok(net) and net.shape[0] == 1:
            net = _mlp2(self.c1, _MaskedGatherFn.apply(net[0], ix, jx)[None], residual=net)     # net + c1(.): the sum in the last GEMM's epilogue
            net = _mlp2(self.c2, _MaskedGatherFn.apply(net[0], jx, ix)[None], residual=net)
        else:
            net = net + self.c1(mask_ix * torch.index_select(net, 1, ix.clamp(min=0)))
            net = net + self.c2(mask_jx * torch.index_select(net, 1, jx.clamp(min=0)))
        net = net + self.agg_kk(net, kk, gk)
        net = net + self.agg_ij(net, None, gp)          # groups of ii * 12345 + jj (enet.py:94)
        net = self.gru(net)
        return net, (self.d(net), self.w(net), None)

    def _torch_groups(self, ii, jj, kk):
        """neighbour indices and torch.unique inverse maps of the two aggregations, cached per graph like `_tables` (a training
        step runs 18 iterations on one graph: 18 neighbour searches, 36 sorts + host reads otherwise)"""
        key = (ii.data_ptr(), jj.data_ptr(), kk.data_ptr(), ii._version, jj._version, kk._version, ii.numel(), jj.numel(), kk.numel())
        hit = getattr(self, "_tg", None)
        if hit is None or hit[0] != key:
            out = []
            for k in (kk, ii * 12345 + jj):
                _, inv = torch.unique(k, return_inverse=True)
                out.append((inv, int(inv.max()) + 1 if inv.numel() else 0))
            out.insert(0, tuple(cuda_ba.neighbors(kk, jj)))
            hit = (key, out, (ii, jj, kk))                 # (the references keep the key's storages alive)
            self._tg = hit
        return hit[1]

    # ------------------------------------------------------------------------------------------ HIP inference path
    def _tables(self, ii, jj, kk):
        """Neighbour / group tables of the current graph, cached.  The cache key is (storage address, version counter, size) of
        ii / jj / kk AND the cache keeps references to those tensors: while they are referenced their storage cannot be freed
        and handed to another tensor, so an equal key means the very same storage, and in-place edits bump the version
        counter.  (DEVO rebuilds ii / jj / kk with torch.cat every frame: fresh tensors -> rebuilt tables.)"""
        key = (ii.data_ptr(), jj.data_ptr(), kk.data_ptr(), ii._version, jj._version, kk._version, ii.numel(), jj.numel(), kk.numel())
        if key != self._graph_key and ii.numel() and _GRAPH_TABLES:
            # one library call (round 6, devo_upd_graph_tables): the patch groups, the neighbours read off them, the pair key and its groups
            ix, jx, tk, tp = cuda_ba.graph_tables(ii, jj, kk)
            E = kk.numel()
            self._graph = (ix, jx, _Groups.from_tables(tk, E), _Groups.from_tables(tp, E))
            self._graph_key = key
            self._graph_refs = (ii, jj, kk)
        elif key != self._graph_key:
            ix, jx = cuda_ba.neighbors(kk, jj)
            il, jl = ii.long(), jj.long()
            # frame-pair key compacted to the window of live frames ON THE DEVICE (round 6: the bounds never visit the host — this runs once per
            # graph, i.e. once per frame in DEVO's steady state, and the .tolist() stalled the eager pipeline): (ii - min ii) * span + (jj - min jj)
            # with span = max jj - min jj + 1 as a 0-d tensor — the same groups as ii * 12345 + jj (enet.py:94), keys in a range of
            # (frames in the window)^2, which is what the index kernels' flag / scan passes walk
            if ii.numel():
                jmin = jl.min()
                pair = ((il - il.min()) * (jl.max() - jmin + 1) + (jl - jmin)).contiguous()
            else:
                pair = il
            self._graph = (ix, jx, _Groups(kk.long().contiguous()), _Groups(pair))
            self._graph_key = key
            self._graph_refs = (ii, jj, kk)                                  # keep the key's storages alive (see above)
        return self._graph

    def invalidate(self):
        """Drop the cached graph tables (e.g. after editing ii / jj / kk through a raw pointer) and the cached weight images."""
        self._graph_key, self._graph, self._graph_refs = None, None, None
        self.invalidate_weights()

    def _rs_plan(self):
        """Everything the fp16 row-resident path hands to its ten launches — weight images, bias / LayerNorm vectors as raw pointers, the
        epsilons — collected once per version of the parameters: the eager call is host-bound otherwise (39 nn.Sequential lookups, 120
        parameter lookups and 60 pointer conversions per call: 185 us where the GPU needs 230)."""
        key = self._pkey()
        pl = self.__dict__.get("_rsplan")
        if pl is not None and pl["key"] == key:
            return pl
        P, img, keep = (lambda t: t.data_ptr()), _rs_image, []       # (plain integers: ctypes takes them for void pointers)

        def im(w):
            t = img(w.detach() if w.requires_grad else w)
            keep.append(t)
            return P(t)
        cat = lambda name, a, b: self._cat(name, a, b)
        Wkk, bkk = cat("agg_kk", self.agg_kk.f, self.agg_kk.g)
        Wij, bij = cat("agg_ij", self.agg_ij.f, self.agg_ij.g)
        W1, b1 = cat("gru1", self.gru[1].gate[0], self.gru[1].res[0])
        W3, b3 = cat("gru3", self.gru[3].gate[0], self.gru[3].res[0])
        keep += [Wkk, bkk, Wij, bij, W1, b1, W3, b3]
        co, g = self.corr, self.gru
        pl = dict(key=key, keep=keep,
                  corr=(im(co[0].weight), P(co[0].bias), im(co[2].weight), P(co[2].bias), P(co[3].weight), P(co[3].bias), float(co[3].eps),
                        im(co[5].weight), P(co[5].bias)),
                  norm=(P(self.norm.weight), P(self.norm.bias), float(self.norm.eps)),
                  c1=(im(self.c1[0].weight), P(self.c1[0].bias), im(self.c1[2].weight), P(self.c1[2].bias)),
                  c2=(im(self.c2[0].weight), P(self.c2[0].bias), im(self.c2[2].weight), P(self.c2[2].bias)),
                  fg_kk=(im(Wkk), P(bkk)), fg_ij=(im(Wij), P(bij)),
                  h_kk=(im(self.agg_kk.h.weight), P(self.agg_kk.h.bias)), h_ij=(im(self.agg_ij.h.weight), P(self.agg_ij.h.bias)),
                  gru=(P(g[0].weight), P(g[0].bias), float(g[0].eps), im(W1), P(b1), im(g[1].res[2].weight), P(g[1].res[2].bias), P(g[2].weight),
                       P(g[2].bias), float(g[2].eps), im(W3), P(b3), im(g[3].res[2].weight), P(g[3].res[2].bias), P(self.d[1].weight), P(self.d[1].bias),
                       P(self.w[1].weight), P(self.w[1].bias)))
        ok = all(t.data_ptr() % 8 == 0 for t in (self.c1[0].bias, self.c1[2].bias, self.c2[0].bias, self.c2[2].bias, bkk, bij)) and \
            all(t.data_ptr() % 16 == 0 for t in (co[3].weight, co[3].bias, self.norm.weight, self.norm.bias, g[0].weight, g[0].bias, g[2].weight, g[2].bias,
                                                  self.d[1].weight, self.w[1].weight))
        pl["ok"] = ok
        self.__dict__["_rsplan"] = pl
        return pl

    def __getstate__(self):
        """(copy.deepcopy, torch.save of the module: the per-version caches stay behind — raw pointers, a half copy of the operator)"""
        d = self.__dict__.copy()
        for k in ("_rsplan", "_shadow", "_params"):
            d.pop(k, None)
        return d

    def _plist(self):
        pl = self.__dict__.get("_params")
        ep = self.__dict__["_epoch"][0]
        if pl is None or pl[0] != ep:                                          # (a replaced Parameter object: the list is re-read from the module tree)
            pl = self.__dict__["_params"] = (ep, list(self.parameters()))
        return pl[1]

    def _pkey(self):
        """What every cached plan / shadow of this operator is keyed on: the epoch of Parameter replacements and (storage address, version
        counter) of every parameter — ~9 us for the 50 parameters (the sum of the versions alone missed `.data = ...`, a
        .float() / .half() round trip and replaced Parameters, leaving raw pointers of freed storage in the plan)."""
        return (self.__dict__["_epoch"][0], tuple((p.data_ptr(), p._version) for p in self._plist()))

    def _apply(self, fn, recurse=True):
        """.to() / .half() / .float() / .cuda(): the parameters' storage moves — no cached pointer survives."""
        out = super()._apply(fn, recurse)
        self.invalidate_weights()
        return out

    def _forward_rs(self, x, inp2, c, ix, jx, Gkk, Gij, E, net32=False):
        """The fp16 operator on the row-resident kernels (csrc/gemm_rs.hip), ten launches: the correlation branch + norm | c1 | c2 + agg_kk's
        f | g | SoftAgg | h | expand-add + agg_ij's f | g | SoftAgg | h | both LayerNorms, both GatedResiduals and the heads.
        net32: `x` (the recurrent state) is fp32 and the new state is returned in fp32 — the call under autocast (devo.py:311), without a
        conversion pass in front of the first and behind the last launch."""
        pl = self._rs_plan()
        if not pl["ok"]:
            return None
        lib, st, P, dim, dt, dev = L.lib(), L.stream(), L.ptr, 384, torch.float16, x.device
        chk = L.check
        out = torch.empty(E, dim, dtype=dt, device=dev)
        corr_fn = lib.devo_upd_rs_corr_f16_net32 if net32 else lib.devo_upd_rs_corr_f16
        chk(corr_fn(P(c), c.stride(0), c.shape[1], *pl["corr"], P(x), P(inp2), *pl["norm"], P(out), E, st), "update.rs_corr_f16")
        x = out
        y = torch.empty_like(x)
        chk(lib.devo_upd_rs_mlp2_fg_f16(P(x), dim, E, P(ix), *pl["c1"], P(x), P(y), E, None, None, None, st), "update.rs_mlp2_f16")
        x = y
        y, fg = torch.empty_like(x), torch.empty(E, 2 * dim, dtype=dt, device=dev)
        chk(lib.devo_upd_rs_mlp2_fg_f16(P(x), dim, E, P(jx), *pl["c2"], P(x), P(y), E, *pl["fg_kk"], P(fg), st), "update.rs_mlp2_f16")
        x = y
        code = L.dtype_code(x)

        def agg(G, h, hint):
            # (a graph seen for the first time: the group count may still be on its way to the host — rows() does not wait for it; `hint` = the
            #  usual rows per group of this aggregation, which only picks the kernel's workgroup shape)
            rows = G.rows()
            ys = torch.empty(rows, dim, dtype=dt, device=dev)
            chk(lib.devo_upd_softagg_hint(P(fg), P(fg[:, dim:]), 2 * dim, P(G.perm), P(G.seg_start), P(G.n_seg_dev), P(ys), P(G.group_of), E, dim, code,
                                          (E // max(rows, 1)) if rows < E else hint, st), "update.softagg")
            hy = torch.empty(rows, dim, dtype=dt, device=dev)
            chk(lib.devo_upd_rs_linear_f16(P(ys), dim, h[0], h[1], None, P(hy), dim, rows, dim, dim, dim, st), "update.rs_linear_f16")
            return hy
        hy = agg(Gkk, pl["h_kk"], 16)
        chk(lib.devo_upd_rs_expand_fg_f16(P(x), P(hy), P(Gkk.group_of), *pl["fg_ij"], P(fg), E, st), "update.rs_expand_fg_f16")
        hy = agg(Gij, pl["h_ij"], 96)
        net_out = torch.empty(E, dim, dtype=torch.float32 if net32 else dt, device=dev)
        dw = torch.empty(2, E, 2, dtype=dt, device=dev)
        g = pl["gru"]
        gru_fn = lib.devo_upd_rs_gru_f16_out32 if net32 else lib.devo_upd_rs_gru_f16
        chk(gru_fn(P(x), P(hy), P(Gij.group_of), g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], g[8], g[9], g[10], g[11], g[12], g[13],
                                    g[14], g[15], g[16], g[17], P(net_out), P(dw[0]), P(dw[1]), E, st), "update.rs_gru_f16")
        return net_out.view(1, E, dim), (dw[0].view(1, E, 2), dw[1].view(1, E, 2), None)

    def _forward_mixed(self, net, inp, corr, flow, ii, jj, kk):
        """The fp16 operator for a caller that keeps the recurrent state in fp32 (devo.py:311 under autocast) -> net fp32, (delta, weight fp16):
        on the row-resident kernels the first launch reads the fp32 state (it enters net + inp + corr unrounded) and the last one writes the new
        state in fp32 — two conversion passes over [E, 384] less per call (34 us of the 45 312-edge steady-state frame); anything else:
        the fp16 operator between explicit conversions."""
        B, E, dim = net.shape
        if (B == 1 and E > 0 and MIXED_STATE and net.dtype == torch.float32 and net.is_cuda and RS_CHAINS and RS_GEMM and dim == 384
                and self.norm.weight.dtype == torch.float16):
            x = net.reshape(E, dim)
            x = x if x.is_contiguous() else x.contiguous()
            inp2, c = inp.reshape(E, dim).half().contiguous(), corr.reshape(E, -1).half()
            if (768 < c.shape[1] <= 896 and c.stride(1) == 1 and c.stride(0) % 2 == 0 and c.data_ptr() % 4 == 0 and inp2.data_ptr() % 16 == 0
                    and x.data_ptr() % 16 == 0):
                L.require_gpu(net, inp, corr, ii, jj, kk)
                ix, jx, Gkk, Gij = self._tables(ii, jj, kk)
                fast = self._forward_rs(x, inp2, c, ix, jx, Gkk, Gij, E, net32=True)
                if fast is not None:
                    return fast
        n16, (d16, w16, _) = self(net.half(), inp.half(), corr.half(), flow, ii, jj, kk)
        return n16.float(), (d16, w16, None)

    def _half_shadow(self):
        """An fp16 copy of this operator (for calls under autocast), rebuilt when a parameter's version counter moves; `.data` edits: invalidate_weights()."""
        key = self._pkey()
        sh = self.__dict__.get("_shadow")
        if sh is None or sh[0] != key:
            m = Update(int(round((self.corr[0].in_features / 98.0) ** 0.5)), self.dim).to(self.norm.weight.device).half().eval()
            with torch.no_grad():
                m.load_state_dict({k: v.detach().half() for k, v in self.state_dict().items()})
            sh = (key, m)
            self.__dict__["_shadow"] = sh
        return sh[1]

    def invalidate_weights(self):
        """Forget the cached operand images of this operator's weights (split / packed / concatenated).  They follow the parameters'
        version counters (copy_, load_state_dict, optimiser steps); an edit through `.data` is invisible to those — call this after one."""
        if getattr(self, "_wcat", None) is not None:
            self._wcat.clear()
        self.__dict__.pop("_shadow", None)
        self.__dict__.pop("_rsplan", None)
        self.__dict__.pop("_params", None)
        invalidate_weight_images()

    def train(self, mode=True):
        self.invalidate_weights()                                            # (mode switches are where checkpoints / EMA weights get swapped in)
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_weights()
        return out

    # The f and g layers of a SoftAgg share their input: ONE GEMM on concatenated weights (cached, rebuilt when a parameter
    # changes).
    def _cat(self, name, a, b):
        key = (a.weight.data_ptr(), a.weight._version, b.weight.data_ptr(), b.weight._version, a.weight.dtype)
        hit = self._wcat.get(name)
        if hit is None or hit[0] != key:
            hit = (key, torch.cat([a.weight, b.weight], 0).contiguous(), torch.cat([a.bias, b.bias], 0).contiguous())
            self._wcat[name] = hit
        return hit[1], hit[2]

    @staticmethod
    def _lin(x, w, b, relu=False, relu_from=None, residual=None):
        """act(x W^T + b) [+ residual, in place]: fp32 rows on the fp16 matrix cores with exact hi + lo splits (csrc/linear.hip, half the
        library's fp32 GEMM time), everything else as a library GEMM (hipBLASLt, bias and ReLU in its epilogue)"""
        if x.dtype == torch.float32 and w.dtype == torch.float32 and _split_ok(x, w.shape[0], w.shape[1]):
            return _linear_split(x, w, b, relu=relu, relu_from=relu_from, residual=residual, out=residual, rs=True)
        # fp16: measured at 21 600 rows against hipBLASLt — 384 outputs 14.4 / 15.6 us, 882 inputs 26.6 / 33.9, 768 outputs 25.3 / 20.1: the
        # wide layers stay with the library unless the launch carries a fusion the library has not (ReLU from a column on, the residual sum)
        if x.dtype == torch.float16 and w.dtype == torch.float16 and (b is None or b.dtype == torch.float16) and _f16_ok(x, w.shape[0], w.shape[1]) \
                and (w.shape[0] <= 384 or relu_from is not None or residual is not None):
            return _linear_f16(x, w, b, relu=relu, relu_from=relu_from, residual=residual, out=residual)
        y = torch._addmm_activation(b, x, w.t(), use_gelu=False) if relu else F.linear(x, w, b)
        if relu_from is not None:
            y[:, relu_from:].relu_()
        return y if residual is None else residual.add_(y)

    def _soft_agg(self, name, agg, net, G, fg=None):
        """-> (h(y), group_of): the aggregated rows and the edge -> group map; the caller adds h(y)[group_of] to net
        (devo_upd_expand_add, or fused into the LayerNorm that follows).  fg: the f | g layer's output when the kernel in front has made it."""
        if fg is None:
            W, b = self._cat(name, agg.f, agg.g)
            fg = self._lin(net, W, b)                              # [E, 2 dim]: f | g
        E, dim = net.shape
        rows = G.rows()                                            # (never waits for a new graph's group count: _Groups.rows)
        y = torch.empty(rows, dim, dtype=net.dtype, device=net.device)
        L.check(L.lib().devo_upd_softagg_hint(L.ptr(fg), L.ptr(fg[:, dim:]), 2 * dim, L.ptr(G.perm), L.ptr(G.seg_start), L.ptr(G.n_seg_dev),
                                              L.ptr(y), L.ptr(G.group_of), E, dim, L.dtype_code(net),
                                              int(E // max(rows, 1)) if rows < E else (16 if name == "agg_kk" else 96), L.stream()),
                "update.softagg")
        return F.linear(y, agg.h.weight, agg.h.bias), G.group_of

    def _gate_res(self, name, gr, x):
        """the three Linear layers of a GatedResidual -> (gate pre-activation, res); the caller fuses x + sigmoid(gate) * res.
        gate | res[0] share their input: one GEMM on concatenated weights, the ReLU from res[0]'s first column on"""
        dim = x.shape[1]
        if (x.dtype == torch.float32 and _split_ok(x, 2 * dim, dim)) or (x.dtype == torch.float16 and _f16_ok(x, 2 * dim, dim)):
            W, b = self._cat(name, gr.gate[0], gr.res[0])
            gr0 = self._lin(x, W, b, relu_from=dim)                # [E, 2 dim]: gate | relu(res[0])
            return gr0[:, :dim], self._lin(gr0[:, dim:], gr.res[2].weight, gr.res[2].bias)
        gate = F.linear(x, gr.gate[0].weight, gr.gate[0].bias)
        return gate, F.linear(self._lin(x, gr.res[0].weight, gr.res[0].bias, relu=True), gr.res[2].weight, gr.res[2].bias)

    def forward(self, net, inp, corr, flow, ii, jj, kk):
        """update operator (enet.py:80): -> net, (delta, weight, None)"""
        if torch.is_grad_enabled() and (net.requires_grad or inp.requires_grad or corr.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            return self.forward_torch(net, inp, corr, ii, jj, kk)
        L.require_gpu(net, inp, corr, ii, jj, kk)
        B, E, dim = net.shape
        if B != 1:
            # enet.py:80-99 is written for a batch; train.py and devo.py always pass ONE sequence.  Batches run entry by entry on the same graph
            # tables (ii / jj / kk are shared by the batch, as in the reference): the kernels' row count is what matters, not the launch count
            if B == 0:
                raise RuntimeError("Update: empty batch")
            outs = [self.forward(net[b:b + 1], inp[b:b + 1], corr[b:b + 1], flow, ii, jj, kk) for b in range(B)]
            return torch.cat([o[0] for o in outs], 0), (torch.cat([o[1][0] for o in outs], 0), torch.cat([o[1][1] for o in outs], 0), None)
        dt = self.norm.weight.dtype
        if (AUTOCAST_F16 and dt == torch.float32 and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.float16 and dim == 384):
            # devo.py:311 calls the fp32 operator under autocast: its Linear layers then run in fp16 (fp32 accumulation), its LayerNorms in fp32.
            # Here that call takes the fp16-storage operator on a half copy of the parameters (kept per parameter version) — the same
            # precision class (every layer output rounded to fp16; the statistics, gates and sums in fp32), a third of the fp32 path's time —
            # and returns what autocast returns: net in fp32, delta / weight in fp16.
            with torch.autocast("cuda", enabled=False):
                return self._half_shadow()._forward_mixed(net, inp, corr, flow, ii, jj, kk)
        if torch.is_autocast_enabled():
            # (the kernels below take the parameters' dtype from first to last: no autocast inside — the library layers among them would
            #  hand fp16 rows to fp32 kernels)
            with torch.autocast("cuda", enabled=False):
                return self.forward(net, inp, corr, flow, ii, jj, kk)
        x, inp2, c = net.reshape(E, dim).to(dt).contiguous(), inp.reshape(E, dim).to(dt).contiguous(), corr.reshape(E, -1).to(dt)
        lib, code = L.lib(), L.dtype_code(x)
        ix, jx, Gkk, Gij = self._tables(ii, jj, kk)
        if (RS_CHAINS and RS_GEMM and dt == torch.float16 and dim == 384 and 768 < c.shape[1] <= 896 and c.stride(1) == 1 and c.stride(0) % 2 == 0
                and c.data_ptr() % 4 == 0 and inp2.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0):
            fast = self._forward_rs(x, inp2, c, ix, jx, Gkk, Gij, E)
            if fast is not None:
                return fast

        # corr MLP (enet.py:59-66) and net = norm(net + inp + corr)  (:82-83), the two adds fused into the LayerNorm
        if (RS_CHAINS and RS_GEMM and dt == torch.float16 and dim == 384 and 768 < c.shape[1] <= 896 and c.stride(1) == 1 and c.stride(0) % 2 == 0
                and c.data_ptr() % 4 == 0 and inp2.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0):
            # the whole correlation branch and norm(net + inp + corr) in ONE launch, rows in LDS (csrc/gemm_rs.hip)
            out = torch.empty_like(x)
            L.check(lib.devo_upd_rs_corr_f16(L.ptr(c), c.stride(0), c.shape[1], L.ptr(_rs_image(self.corr[0].weight.detach())), L.ptr(self.corr[0].bias),
                                             L.ptr(_rs_image(self.corr[2].weight.detach())), L.ptr(self.corr[2].bias), L.ptr(self.corr[3].weight),
                                             L.ptr(self.corr[3].bias), float(self.corr[3].eps), L.ptr(_rs_image(self.corr[5].weight.detach())),
                                             L.ptr(self.corr[5].bias), L.ptr(x), L.ptr(inp2), L.ptr(self.norm.weight), L.ptr(self.norm.bias),
                                             float(self.norm.eps), L.ptr(out), E, L.stream()), "update.rs_corr_f16")
            x, c = out, None
        elif _mlp2_ok(c, self.corr[0], self.corr[2]):                 # Linear - ReLU - Linear: one launch, the intermediate stays in LDS
            c = _mlp2_f16(c, self.corr[0], self.corr[2])
        else:
            c = self._lin(c, self.corr[0].weight, self.corr[0].bias, relu=True)
            c = self._lin(c, self.corr[2].weight, self.corr[2].bias)
        if c is not None:
            c = _ln(c, self.corr[3], relu=True)
            c = self._lin(c, self.corr[5].weight, self.corr[5].bias)
            x = _ln(x, self.norm, add1=inp2, add2=c)

        # neighbour mixing along the patch trajectory (:86-91)
        tails = RS_CHAINS and RS_GEMM and dt == torch.float16 and dim == 384     # the SoftAgg's f | g layers ride on the kernels in front of them
        fg_kk = None
        for mlp, idx in ((self.c1, ix), (self.c2, jx)):
            if _mlp2_ok(x, mlp[0], mlp[2]) and x.is_contiguous():
                if tails and mlp is self.c2:
                    x, fg_kk = _mlp2_f16(x, mlp[0], mlp[2], residual=x, gather=idx, fg=self._cat("agg_kk", self.agg_kk.f, self.agg_kk.g))
                    continue
                x = _mlp2_f16(x, mlp[0], mlp[2], residual=x, gather=idx)     # net + c(mask * net[:, idx]): gather, both layers and the sum in one launch
                continue
            t = torch.empty_like(x)
            L.check(lib.devo_upd_masked_gather(L.ptr(x), L.ptr(idx), L.ptr(t), E, dim, code, L.stream()), "update.masked_gather")
            t = self._lin(t, mlp[0].weight, mlp[0].bias, relu=True)
            self._lin(t, mlp[2].weight, mlp[2].bias, residual=x)     # x += c(t)

        # soft aggregation over the edges of a patch, then over the edges of a frame pair (:93-94); the second expand is
        # fused into the LayerNorm of the "gru" (:52-57), and each GatedResidual into the op that consumes it
        hy, grp = self._soft_agg("agg_kk", self.agg_kk, x, Gkk, fg=fg_kk)
        if tails and x.is_contiguous() and hy.is_contiguous() and x.data_ptr() % 16 == 0:
            Wij, bij = self._cat("agg_ij", self.agg_ij.f, self.agg_ij.g)     # x += hy[grp] and agg_ij's f | g layer on the same rows: one launch
            fg_ij = torch.empty(E, 2 * dim, dtype=dt, device=x.device)
            L.check(lib.devo_upd_rs_expand_fg_f16(L.ptr(x), L.ptr(hy), L.ptr(grp), L.ptr(_rs_image(Wij)), L.ptr(bij), L.ptr(fg_ij), E, L.stream()),
                    "update.rs_expand_fg_f16")
            hy, grp = self._soft_agg("agg_ij", self.agg_ij, x, Gij, fg=fg_ij)
        else:
            L.check(lib.devo_upd_expand_add(L.ptr(x), L.ptr(hy), L.ptr(grp), E, dim, code, L.stream()), "update.expand_add")
            hy, grp = self._soft_agg("agg_ij", self.agg_ij, x, Gij)
        if RS_CHAINS and dt == torch.float16 and dim == 384 and RS_GEMM and x.is_contiguous() and hy.is_contiguous():
            # everything behind the aggregation is row-local: both LayerNorms, both GatedResiduals and the heads in ONE launch, rows in LDS
            net_out = torch.empty_like(x)
            delta = torch.empty(E, 2, dtype=dt, device=x.device)
            weight = torch.empty(E, 2, dtype=dt, device=x.device)
            W1, b1 = self._cat("gru1", self.gru[1].gate[0], self.gru[1].res[0])
            W3, b3 = self._cat("gru3", self.gru[3].gate[0], self.gru[3].res[0])
            r1, r3 = self.gru[1].res[2], self.gru[3].res[2]
            L.check(lib.devo_upd_rs_gru_f16(L.ptr(x), L.ptr(hy), L.ptr(grp), L.ptr(self.gru[0].weight), L.ptr(self.gru[0].bias), float(self.gru[0].eps),
                                            L.ptr(_rs_image(W1)), L.ptr(b1), L.ptr(_rs_image(r1.weight.detach())), L.ptr(r1.bias), L.ptr(self.gru[2].weight),
                                            L.ptr(self.gru[2].bias), float(self.gru[2].eps), L.ptr(_rs_image(W3)), L.ptr(b3),
                                            L.ptr(_rs_image(r3.weight.detach())), L.ptr(r3.bias), L.ptr(self.d[1].weight), L.ptr(self.d[1].bias),
                                            L.ptr(self.w[1].weight), L.ptr(self.w[1].bias), L.ptr(net_out), L.ptr(delta), L.ptr(weight), E, L.stream()),
                    "update.rs_gru_f16")
            return net_out.view(1, E, dim), (delta.view(1, E, 2), weight.view(1, E, 2), None)
        x = _ln(x, self.gru[0], expand=(hy, grp))                                  # LN(net + agg_ij(net))
        x = _ln(x, self.gru[2], gated=self._gate_res("gru1", self.gru[1], x))     # LN(GatedResidual(.))
        gate, res = self._gate_res("gru3", self.gru[3], x)

        net_out = torch.empty_like(x)
        delta = torch.empty(E, 2, dtype=dt, device=x.device)
        weight = torch.empty(E, 2, dtype=dt, device=x.device)
        L.check(lib.devo_upd_heads(L.ptr(x), L.ptr(gate), gate.stride(0), L.ptr(res), L.ptr(net_out), L.ptr(self.d[1].weight),
                                   L.ptr(self.d[1].bias), L.ptr(self.w[1].weight), L.ptr(self.w[1].bias), L.ptr(delta), L.ptr(weight),
                                   E, dim, code, L.stream()), "update.heads")
        x = net_out
        return x.view(1, E, dim), (delta.view(1, E, 2), weight.view(1, E, 2), None)
