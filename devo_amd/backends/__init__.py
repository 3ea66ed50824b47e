"""Drop-in replacements for the reference's three pybind11 extension modules.

The reference imports them by their top-level names (devo/altcorr/correlation.py:2 `import cuda_corr`,
devo/fastba/ba.py:2 `import cuda_ba`, devo/lietorch/group_ops.py:1 `import lietorch_backends`).
`install()` registers the HIP-backed modules under exactly those names, so an unmodified checkout of the
reference picks them up:

    import devo_amd.backends as b; b.install()
    import devo            # the reference package, unchanged
"""
import os
import sys

_native = False          # False: not looked for yet; None: absent / switched off


def native():
    """devo_amd._C — the compiled binding (csrc/bind.cpp: pybind11 sub-modules cuda_corr / cuda_ba / lietorch_backends taking torch::Tensor,
    + torch.ops.devo_hip) — or None: not built (python -m devo_amd.build), DEVO_BINDING=ctypes, or DEVO_LIB names another build of the
    library (the compiled binding is linked against devo_amd/lib/libdevo_hip.so).  The ctypes modules of this package are the no-compile
    form of the same binding; both call the same C ABI, neither has a CPU path."""
    global _native
    if _native is False:
        _native = None
        if os.environ.get("DEVO_BINDING", "native") != "ctypes" and not os.environ.get("DEVO_LIB"):
            try:
                from .. import _C
                from .. import _lib
                if _C.abi_version() == _lib.ABI_VERSION:
                    _native = _C
            except ImportError:
                pass
    return _native


def install():
    """Register the three modules under the names the reference imports.  With the compiled binding present these are ITS sub-modules
    (the reference's exact signatures, torch::Tensor in, ATen allocations, c10::hip::getCurrentHIPStream()); else the ctypes modules."""
    from . import cuda_corr, cuda_ba, lietorch_backends
    n = native()
    mods = (n.cuda_corr, n.cuda_ba, n.lietorch_backends) if n is not None else (cuda_corr, cuda_ba, lietorch_backends)
    sys.modules["cuda_corr"], sys.modules["cuda_ba"], sys.modules["lietorch_backends"] = mods
    from . import ring
    ring.track_ring_writes(True)         # per-slot maintenance of the converted ring buffers (ring.py; both bindings; DEVO_RING_SLOTS=0: off)
    return mods
