"""`devo.fastba` as the rest of DEVO imports it (devo/devo.py:13, devo/enet.py:11): three callables over the HIP bundle
adjustment.  The reference module binds them straight to its CUDA extension (devo/fastba/ba.py:4-8); here they go
through devo_amd.backends.cuda_ba and keep one BA workspace alive between calls of the same size, so that the update
loop does not allocate on every iteration."""
from .backends import cuda_ba as _ba

_ws_cache = {}


def _workspace(n_edges, n_patch_slots, n_opt, device):
    key = (int(n_edges), int(n_patch_slots), int(n_opt), str(device))
    ws = _ws_cache.get(key)
    if ws is None:
        _ws_cache.clear()                                   # one live workspace: the graph size changes once per frame
        ws = _ws_cache[key] = _ba.workspace(*key[:3], device)
    return ws


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2):
    """Gauss-Newton bundle adjustment, IN PLACE on the storage of `poses` (an SE3 object or a tensor) and `patches`;
    returns [] like the extension (callers rely on the mutation, devo/devo.py:337)."""
    pose_data = getattr(poses, "data", poses)
    P = patches.shape[-1]
    ws = _workspace(ii.numel(), patches.numel() // (3 * P * P), int(t1) - int(t0), pose_data.device)
    return _ba.forward(pose_data, patches, intrinsics, target, weight, lmbda, ii, jj, kk, int(t0), int(t1), int(iterations), ws=ws)


def neighbors(ii, jj):
    """[ix, jx]: previous / next edge of the same patch in frame order, -1 where there is none (ba.cpp:104-149)."""
    return _ba.neighbors(ii, jj)


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """coords [1, E, 2, P, P] of every patch pixel in its target frame (ba_cuda.cu:368-418)."""
    return _ba.reproject(poses, patches, intrinsics, ii, jj, kk)
