"""Host-side mirror of devo/fastba/ba.py:4-8."""
from .backends import cuda_ba

neighbors = cuda_ba.neighbors
reproject = cuda_ba.reproject


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2):
    """In-place bundle adjustment on poses.data / patches (devo/fastba/ba.py:7-8)."""
    return cuda_ba.forward(poses.data, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations)
