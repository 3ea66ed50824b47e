"""CPU checks of the drop-in boundary: the C-ABI library builds/loads without a GPU and exports every symbol
include/devo_hip.h declares; the Python layer refuses CPU tensors (no fallback)."""
import ctypes
import os
import re
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "devo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(devo_[a-zA-Z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def libpath():
    from devo_amd import build
    return build.build_lib(verbose=False)


def test_header_symbols_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/devo_hip.h but not exported"
    m = re.search(r"#define\s+DEVO_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "devo_hip.h")).read())
    from devo_amd import _lib
    assert lib.devo_abi_version() == int(m.group(1)) == _lib.ABI_VERSION      # header, library and ctypes table agree


def test_stale_library_is_refused(libpath, monkeypatch):
    """A library of another ABI version (argument lists differ between versions) must not be bound."""
    from devo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.lib()


def test_python_binding_covers_header(libpath):
    from devo_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == _declared()
    _lib.lib()      # sets argtypes for every symbol; raises if one is missing


def test_workspace_queries(libpath):
    from devo_amd import _lib
    L = _lib.lib()
    assert L.devo_ba_workspace_bytes(21600, 1440, 14) > 21600 * 12 * 4
    assert L.devo_ba_workspace_bytes(100, 10, 33) > 199 * 199 * 4     # more than 32 optimised poses: the system in global memory
    assert L.devo_ba_workspace_bytes(100, 10, 129) == 0               # more than 128: unsupported
    assert L.devo_neighbors_workspace_bytes(21600) > 21600 * 8


def test_no_cpu_fallback(libpath):
    from devo_amd.backends import cuda_corr, cuda_ba, lietorch_backends
    x = torch.zeros(4, 7)
    x[:, 6] = 1
    with pytest.raises(RuntimeError):
        lietorch_backends.inv(3, x)
    with pytest.raises(RuntimeError):
        cuda_ba.neighbors(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))
    with pytest.raises(RuntimeError):
        cuda_corr.patchify_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 2, 2), 1)


def test_dropin_install(libpath):
    import sys
    import devo_amd.backends as b
    mods = b.install()
    try:
        import cuda_corr, cuda_ba, lietorch_backends  # noqa: F401  (the names the reference imports)
        for name in ("forward", "backward", "patchify_forward", "patchify_backward"):
            assert callable(getattr(cuda_corr, name))                   # correlation.cpp:58-62
        for name in ("forward", "neighbors", "reproject"):
            assert callable(getattr(cuda_ba, name))                     # ba.cpp:153-155
        for name in ("expm", "logm", "inv", "mul", "adj", "adjT", "act", "act4"):
            assert callable(getattr(lietorch_backends, name)) and callable(getattr(lietorch_backends, name + "_backward"))
        for name in ("as_matrix", "projector", "Jinv"):
            assert callable(getattr(lietorch_backends, name))           # lietorch.cpp:312-314
    finally:
        for m in ("cuda_corr", "cuda_ba", "lietorch_backends"):
            sys.modules.pop(m, None)


def test_compiled_binding_has_the_reference_interfaces(libpath):
    """devo_amd._C (csrc/bind.cpp): three pybind11 sub-modules with the reference's function names (correlation.cpp:58-62, ba.cpp:153-155,
    lietorch.cpp:287-314) taking torch.Tensor, registered under torch.ops.devo_hip as well; backends.install() hands out these modules;
    CPU tensors raise (no fallback); the ctypes modules stay importable as the no-compile form (DEVO_BINDING=ctypes)."""
    import inspect
    import sys
    from devo_amd import build, _lib
    build.build_binding(verbose=False)
    import devo_amd.backends as b
    b._native = False
    N = b.native()
    assert N is not None and N.abi_version() == _lib.ABI_VERSION
    for name in ("forward", "backward", "patchify_forward", "patchify_backward"):
        assert callable(getattr(N.cuda_corr, name))
    for name in ("forward", "neighbors", "reproject"):
        assert callable(getattr(N.cuda_ba, name))
    for name in ("expm", "logm", "inv", "mul", "adj", "adjT", "act", "act4"):
        assert callable(getattr(N.lietorch_backends, name)) and callable(getattr(N.lietorch_backends, name + "_backward"))
    for name in ("as_matrix", "projector", "Jinv"):
        assert callable(getattr(N.lietorch_backends, name))
    # the reference's positional signatures (pybind11 docstrings carry them)
    assert N.cuda_corr.forward.__doc__.count("torch.Tensor") == 6 and "arg5" in N.cuda_corr.forward.__doc__ and "arg6" not in N.cuda_corr.forward.__doc__
    assert "iterations" in N.cuda_ba.forward.__doc__ and N.lietorch_backends.inv.__doc__.startswith("inv(arg0") and "arg1: torch.Tensor" in N.lietorch_backends.inv.__doc__
    for op in ("corr_forward", "corr_backward", "patchify_forward", "patchify_backward", "ba_forward", "ba_neighbors", "ba_reproject", "se3_exp", "se3_inv", "se3_mul",
               "se3_act4", "se3_adjT"):
        assert hasattr(torch.ops.devo_hip, op), op
    mods = b.install()
    try:
        import cuda_corr, cuda_ba, lietorch_backends  # noqa: F401
        assert mods[0] is N.cuda_corr and sys.modules["cuda_ba"] is N.cuda_ba and sys.modules["lietorch_backends"] is N.lietorch_backends
        x = torch.zeros(4, 7)
        x[:, 6] = 1
        for call in (lambda: lietorch_backends.inv(3, x), lambda: cuda_ba.neighbors(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long)),
                     lambda: cuda_corr.patchify_forward(torch.zeros(1, 4, 8, 8), torch.zeros(1, 2, 2), 1), lambda: torch.ops.devo_hip.se3_inv(3, x),
                     lambda: lietorch_backends.inv(1, x)):
            with pytest.raises(RuntimeError):
                call()
    finally:
        for m in ("cuda_corr", "cuda_ba", "lietorch_backends"):
            sys.modules.pop(m, None)


def test_ring_write_tracker_is_transparent_on_cpu():
    """devo_amd.backends.ring wraps torch.Tensor.__setitem__ (installed by backends.install() for the per-slot maintenance of the converted ring
    buffers): CPU tensors — and anything the binding holds no converted copy of — behave exactly as before, version counters included; the wrapper
    can be removed again and installing twice is harmless."""
    from devo_amd.backends import ring
    was = ring.tracking()                                          # (an earlier test's install() may have switched it on)
    try:
        ring.track_ring_writes(False)
        base = torch.Tensor.__setitem__
        on = ring.track_ring_writes(True)
        if os.environ.get("DEVO_RING_SLOTS", "1") == "0":
            assert not on
            return
        assert on and ring.tracking() and torch.Tensor.__setitem__ is not base
        assert ring.track_ring_writes(True)                       # idempotent
        a = torch.zeros(1, 4, 3, 5, 5)
        v0 = a._version
        a[:, 1] = torch.ones(1, 3, 5, 5)
        a[0, 2] = a[0, 1]
        a[:, torch.tensor([0, 3])] = 2.0
        assert a._version == v0 + 3 and float(a[0, 1].sum()) == 75.0 and float(a[0, 2].sum()) == 75.0 and float(a[0, 3].sum()) == 150.0
        with pytest.raises((IndexError, RuntimeError)):
            a[:, 9] = 1.0                                          # errors pass through
        assert not ring.track_ring_writes(False) and torch.Tensor.__setitem__ is base
    finally:
        ring.track_ring_writes(was)
