"""ctypes binding of libdevo_hip.so (include/devo_hip.h).

There is NO fallback: if the library is missing or a tensor is not on a GPU the call raises.  PyTorch is
used only for device memory and streams (tensor.data_ptr(), torch.cuda.current_stream()).
"""
import ctypes
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DEVO_LIB=<path>: another build of the library (tools/build_variant.sh: A/B of kernels inside one gpurun call); the compiled binding is linked
# against devo_amd/lib/libdevo_hip.so, so backends.native() steps aside and the ctypes modules carry the calls
LIB_PATH = os.path.abspath(os.environ["DEVO_LIB"]) if os.environ.get("DEVO_LIB") else os.path.join(_HERE, "lib", "libdevo_hip.so")

DEVO_F32, DEVO_F16, DEVO_F64 = 0, 1, 2
ABI_VERSION = 5                # include/devo_hip.h DEVO_ABI_VERSION: a library of another version is refused (argument lists differ)
CBLOCK_SPLIT8 = -8             # DEVO_CBLOCK_SPLIT8: fp32 level in the split-blocked format of devo_corr_pyramid_split
PLAN_TAIL = 4104               # DEVO_CORR_PLAN_TAIL: a plan buffer that can hold a group plan has 2 n + 2 + PLAN_TAIL ints
PLAN_EDGES, PLAN_GROUPS = 0, 1
_DT = {torch.float32: DEVO_F32, torch.float16: DEVO_F16, torch.float64: DEVO_F64}

_c_i64p = ctypes.POINTER(ctypes.c_int64)
_vp, _i, _i64, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_float

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/devo_hip.h exactly
_SIGNATURES = {
    "devo_abi_version": [],
    "devo_last_error": [],
    "devo_stream_capturing": [_vp],
    "devo_corr_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _c_i64p, _i, _i64, _i64, _i64, _i, _i, _vp, ctypes.c_float, _vp, _vp, _vp],
    "devo_corr_forward_pyramid2": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, ctypes.POINTER(ctypes.c_int), _c_i64p,
                                   ctypes.POINTER(ctypes.c_int), _i64, _i64, _c_i64p, _i, _i, _vp, ctypes.POINTER(ctypes.c_float), _vp, _vp, _vp, _i, _vp],
    "devo_corr_patch_operand_bytes": [_i, _i, _i],
    "devo_corr_patch_transpose": [_vp, _vp, _i, _i, _i, _vp],
    "devo_corr_pyramid_split": [_vp, _c_i64p, _i, _i, _i, _i, _i, _vp, _i64, _vp, _vp],
    "devo_corr_pyramid_split_frames": [_vp, _c_i64p, _i, _i, _i, _i, _i, _vp, _i64, _vp, _vp, _vp],
    "devo_corr_patch_transpose_range": [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    "devo_pyramid_build": [_vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i, _vp],
    "devo_corr_order": [_vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.c_float, _i, _i, _i, _vp],
    "devo_corr_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _c_i64p, _i64, _i, _i, _vp, ctypes.c_size_t, _vp],
    "devo_corr_backward_workspace_bytes": [_i, _i, _i, _i, _i, _i, _i],
    "devo_corr_backward_last_path": [],
    "devo_corr_forward_last_path": [],
    "devo_ba_last_path": [],
    "devo_patchify_forward": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _c_i64p, _i, _i, _vp],
    "devo_patchify_backward": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _c_i64p, _i, _i, _vp],
    "devo_ba_workspace_bytes": [_i, _i, _i],
    "devo_ba_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp],
    "devo_ba_prepare": [_vp, _i, _i, _i, _vp, _sz, _vp],
    "devo_ba_prepare_plan": [_vp, _i, _i, _i, _vp, _sz, _vp, _i, _i, _i, _i, _vp],
    "devo_ba_prepared_tables": [_vp, _sz, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "devo_ba_forward_prepared": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp],
    "devo_ba_forward_prepared_delta": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _vp],
    "devo_ba_import_tables": [_vp, _sz, _i, _i, _vp, _sz, _i, _i, _i, _vp],
    "devo_ba_forward_prepared_delta_plan": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp,
                                            _vp, _i, _i, _i, _i, _vp],
    "devo_ba_solve_terms": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_float, _vp, _sz, _vp, _vp, _vp, _vp],
    "devo_ba_solve_terms_backward": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp],
    "devo_ba_apply_step": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp],
    "devo_ba_apply_step_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "devo_transform_vjp": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "devo_ba_edge_terms": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "devo_ba_edge_terms_backward": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "devo_ba_table_offsets": [_i, _i, _i, _vp],
    "devo_upd_graph_tables": [_vp, _vp, _vp, _i, _i, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp],
    "devo_neighbors_workspace_bytes": [_i],
    "devo_ba_neighbors": [_vp, _vp, _vp, _vp, _i, _vp, _sz, _vp],
    "devo_ba_reproject": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "devo_transform": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp],
}
_f = ctypes.c_float
_SIGNATURES.update({
    "devo_instnorm_workspace_bytes": [_i, _i],
    "devo_instnorm_cl": [_vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _sz, _i, _vp],
    "devo_instnorm_bias_cl": [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp, _sz, _i, _vp],
    "devo_bias_act_cl": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "devo_upd_layernorm": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _i, _vp],
    "devo_upd_layernorm_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _vp, _vp],
    "devo_upd_masked_gather": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "devo_upd_softagg": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "devo_upd_softagg_hint": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "devo_upd_softagg_backward": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _vp],
    "devo_upd_expand_add": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "devo_upd_gated_residual": [_vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _vp],
    "devo_upd_gated_residual_backward": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "devo_upd_heads": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp],
    "devo_upd_split_weight_bytes": [_i, _i],
    "devo_upd_split_weight": [_vp, _i64, _i64, _i, _i, _vp, _vp],
    "devo_upd_linear_split": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "devo_upd_pack_weight_f16_bytes": [_i, _i],
    "devo_upd_pack_weight_f16": [_vp, _i64, _i64, _i, _i, _vp, _vp],
    "devo_upd_linear_f16": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "devo_upd_rs_weight_bytes": [_i, _i],
    "devo_upd_rs_supported": [_i, _i],
    "devo_upd_rs_pack_weight_f16": [_vp, _i64, _i64, _i, _i, _vp, _vp],
    "devo_upd_rs_linear_f16": [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "devo_upd_rs_gru_f16": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "devo_upd_rs_gru_f16_out32": [_vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "devo_upd_rs_mlp2_f16": [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "devo_upd_rs_mlp2_fg_f16": [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "devo_upd_rs_expand_fg_f16": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "devo_upd_rs_corr_f16": [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp],
    "devo_upd_rs_corr_f16_net32": [_vp, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp],
    "devo_upd_rs_split_weight_bytes": [_i, _i],
    "devo_upd_rs_split_supported": [_i, _i],
    "devo_upd_rs_split_weight": [_vp, _i64, _i64, _i, _i, _vp, _vp],
    "devo_upd_rs_linear_split": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "devo_upd_mlp2_weight_bytes": [_i],
    "devo_upd_mlp2_pack_weight": [_vp, _i64, _i64, _i, _vp, _vp],
    "devo_upd_mlp2_f16": [_vp, _i64, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i64, _i, _vp],
    "devo_upd_dw_workspace_bytes": [_i, _i, _i],
    "devo_upd_dw_split": [_vp, _i64, _vp, _i64, _i, _i, _i, _vp, _vp, _i64, _vp, _vp],
})
_SIGNATURES.update({
    "devo_voxelize": [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp],
    "devo_voxel_std_workspace_bytes": [_i],
    "devo_voxel_std": [_vp, _i, _i64, _vp, _sz, _vp],
})
for _n in ("exp", "log", "inv"):
    _SIGNATURES[f"devo_se3_{_n}"] = [_vp, _vp, _i64, _i, _vp]
    _SIGNATURES[f"devo_se3_{_n}_backward"] = [_vp, _vp, _vp, _i64, _i, _vp]
for _n in ("mul", "adj", "adjT", "act", "act4"):
    _SIGNATURES[f"devo_se3_{_n}"] = [_vp, _vp, _vp, _i64, _i, _vp]
    _SIGNATURES[f"devo_se3_{_n}_backward"] = [_vp, _vp, _vp, _vp, _vp, _i64, _i, _vp]
_SIGNATURES["devo_se3_as_matrix"] = [_vp, _vp, _i64, _i, _vp]
_SIGNATURES["devo_se3_jinv"] = [_vp, _vp, _vp, _i64, _i, _vp]
_RESTYPE = {"devo_last_error": ctypes.c_char_p, "devo_instnorm_workspace_bytes": _sz, "devo_voxel_std_workspace_bytes": _sz, "devo_ba_workspace_bytes": _sz, "devo_neighbors_workspace_bytes": _sz,
             "devo_corr_backward_workspace_bytes": _sz, "devo_corr_patch_operand_bytes": _sz, "devo_upd_split_weight_bytes": _sz, "devo_upd_dw_workspace_bytes": _sz, "devo_upd_pack_weight_f16_bytes": _sz, "devo_upd_mlp2_weight_bytes": _sz, "devo_upd_rs_weight_bytes": _sz, "devo_upd_rs_split_weight_bytes": _sz}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m devo_amd.build` (hipcc --offload-arch=gfx950). "
                "devo_amd has no CPU fallback.")
        h = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGNATURES.items():
            fn = getattr(h, name)
            fn.argtypes = args
            fn.restype = _RESTYPE.get(name, ctypes.c_int)
        got = h.devo_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} has ABI version {got}, this package binds version {ABI_VERSION}: rebuild it with `python -m devo_amd.build --force`")
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().devo_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"devo_amd: unsupported dtype {t.dtype}")


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("devo_amd: tensors must live on the GPU (the HIP path has no CPU fallback)")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # the raw hipStream_t of the current stream: one C call


def stream():
    """The current stream of the current device as the C ABI wants it.  `torch.cuda.current_stream().cuda_stream` builds a Stream
    object and resolves the device through four Python layers (~9 us; the reference's eager call sequence asks seven times per update
    iteration: bench.py --api reference); torch's own raw accessor does the same in one call."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch._C._cuda_getDevice()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def i64arr(vals):
    return (ctypes.c_int64 * len(vals))(*[int(v) for v in vals])
