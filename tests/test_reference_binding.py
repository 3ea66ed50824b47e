"""The drop-in boundary, checked against the reference's OWN Python wrappers (build container only: /root/reference does not exist on
the GPU box, where this file skips).  After devo_amd.backends.install() the reference's L1 modules — which bind attribute names of
`cuda_corr`, `cuda_ba` and `lietorch_backends` at import time (devo/altcorr/correlation.py:2,11,28,40,47; devo/fastba/ba.py:2-8;
devo/lietorch/group_ops.py:1,28-66) — must import unchanged, every GroupOp must have resolved its backend functions, and a CPU call
must raise the no-fallback error instead of computing anything."""
import importlib
import os
import sys
import types
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "devo")), reason="the reference checkout is not on this machine")

# the names the reference's GroupOp subclasses bind (group_ops.py:28-66): forward ops and the backward ops that exist
FORWARD = ["expm", "logm", "inv", "mul", "adj", "adjT", "act", "act4", "Jinv", "as_matrix"]
BACKWARD = ["expm_backward", "logm_backward", "inv_backward", "mul_backward", "adj_backward", "adjT_backward", "act_backward", "act4_backward"]


@pytest.fixture()
def reference_package():
    """install() + the reference's `devo` package as a bare namespace (its __init__ is empty; devo.enet / devo.devo need torchvision and
    yacs, which this image lacks and the hot path does not touch) + a functional torch_scatter.scatter_sum (devo/ba.py:2; absent on ROCm
    wheels here: plain index_add)."""
    import devo_amd.backends as b
    saved = {k: sys.modules.get(k) for k in list(sys.modules) if k == "devo" or k.startswith("devo.") or k in ("cuda_corr", "cuda_ba", "lietorch_backends", "torch_scatter")}
    for k in saved:
        sys.modules.pop(k, None)
    mods = b.install()
    ts = types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
        dim = dim % src.dim()
        size = list(src.shape)
        size[dim] = int(dim_size) if dim_size is not None else (int(index.max()) + 1 if index.numel() else 0)
        res = torch.zeros(size, dtype=src.dtype, device=src.device) if out is None else out
        return res.index_add_(dim, index, src)
    ts.scatter_sum = scatter_sum
    sys.modules["torch_scatter"] = ts
    sys.path.insert(0, REF)
    try:
        yield mods
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k == "devo" or k.startswith("devo.") or k in ("cuda_corr", "cuda_ba", "lietorch_backends", "torch_scatter")]:
            sys.modules.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v


def test_reference_wrappers_bind_to_the_installed_modules(reference_package):
    cuda_corr, cuda_ba, lietorch_backends = reference_package
    lt = importlib.import_module("devo.lietorch")
    go = importlib.import_module("devo.lietorch.group_ops")
    ac = importlib.import_module("devo.altcorr")
    corr_mod = importlib.import_module("devo.altcorr.correlation")
    fb = importlib.import_module("devo.fastba")
    pops = importlib.import_module("devo.projective_ops")
    ba = importlib.import_module("devo.ba")
    assert os.path.realpath(lt.__file__).startswith(REF) and os.path.realpath(ba.__file__).startswith(REF)

    # every GroupOp subclass resolved its forward / backward op to the HIP-backed module's function: 10 + 8 names (+ projector below)
    ops = {c.__name__: c for c in vars(go).values() if isinstance(c, type) and issubclass(c, go.GroupOp) and c is not go.GroupOp}
    want = {"Exp": ("expm", "expm_backward"), "Log": ("logm", "logm_backward"), "Inv": ("inv", "inv_backward"), "Mul": ("mul", "mul_backward"),
            "Adj": ("adj", "adj_backward"), "AdjT": ("adjT", "adjT_backward"), "Act3": ("act", "act_backward"), "Act4": ("act4", "act4_backward"),
            "Jinv": ("Jinv", None), "ToMatrix": ("as_matrix", None)}
    assert set(ops) == set(want)
    assert sorted(f for f, _ in want.values()) == sorted(FORWARD) and sorted(b_ for _, b_ in want.values() if b_) == sorted(BACKWARD)
    for name, (fwd, bwd) in want.items():
        assert ops[name].forward_op is getattr(lietorch_backends, fwd), name
        assert (ops[name].backward_op is None) if bwd is None else (ops[name].backward_op is getattr(lietorch_backends, bwd)), name
    assert callable(lietorch_backends.projector)                     # lietorch.cpp:313 (19th name; not on the hot path)

    # altcorr / fastba: the wrappers call through to the installed modules' functions
    assert corr_mod.cuda_corr is cuda_corr and fb.ba.cuda_ba is cuda_ba
    assert fb.neighbors is cuda_ba.neighbors and fb.reproject is cuda_ba.reproject
    assert callable(ac.corr) and callable(ac.patchify) and callable(fb.BA)
    assert callable(pops.transform) and callable(ba.BA)


def test_a_cpu_call_through_the_reference_wrappers_raises_instead_of_falling_back(reference_package):
    lt = importlib.import_module("devo.lietorch")
    X = lt.SE3.Identity(1)
    with pytest.raises(RuntimeError, match="GPU"):
        X.inv()
    ac = importlib.import_module("devo.altcorr")
    with pytest.raises(RuntimeError, match="GPU"):
        ac.corr(torch.zeros(1, 2, 8, 3, 3), torch.zeros(1, 2, 8, 4, 4), torch.zeros(1, 3, 2, 3, 3), torch.zeros(3, dtype=torch.long),
                torch.zeros(3, dtype=torch.long), 3)
    fb = importlib.import_module("devo.fastba")
    with pytest.raises(RuntimeError, match="GPU"):
        fb.neighbors(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))
