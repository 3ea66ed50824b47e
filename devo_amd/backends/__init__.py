"""Drop-in replacements for the reference's three pybind11 extension modules.

The reference imports them by their top-level names (devo/altcorr/correlation.py:2 `import cuda_corr`,
devo/fastba/ba.py:2 `import cuda_ba`, devo/lietorch/group_ops.py:1 `import lietorch_backends`).
`install()` registers the HIP-backed modules under exactly those names, so an unmodified checkout of the
reference picks them up:

    import devo_amd.backends as b; b.install()
    import devo            # the reference package, unchanged
"""
import sys


def install():
    from . import cuda_corr, cuda_ba, lietorch_backends
    sys.modules["cuda_corr"] = cuda_corr
    sys.modules["cuda_ba"] = cuda_ba
    sys.modules["lietorch_backends"] = lietorch_backends
    return cuda_corr, cuda_ba, lietorch_backends
