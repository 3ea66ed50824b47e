"""`devo.fastba` as the rest of DEVO imports it (devo/devo.py:13, devo/enet.py:11): three callables over the HIP bundle
adjustment.  The reference module binds them straight to its CUDA extension (devo/fastba/ba.py:4-8); here they go
through devo_amd.backends.cuda_ba and keep one BA workspace alive between calls of the same size, so that the update
loop does not allocate on every iteration."""
import torch
from .backends import cuda_ba as _ba

_ws_cache = {}
_status = {}                  # device -> int32 [1]: 0 = ok, k > 0 = the Cholesky factorisation broke down in iteration k, -1 = bad workspace
_host = {}                    # device -> (pinned int32 [1], event): the status is copied out asynchronously after every BA()
_pending = {}                 # device -> True while the status of the last BA() has not been looked at
MAX_OPTIMISED_POSES = 128     # devo_ba_forward: up to 32 the reduced system lives in one workgroup's LDS, beyond in global memory (slower)


class BAFailure(RuntimeError):
    """The reduced camera system of the previous BA() call was not positive definite (the reference's cuSOLVER call throws
    at this point, ba_cuda.cu:521-523, and devo/devo.py:336-340 catches it and prints "Warning BA failed")."""


def _workspace(n_edges, n_patch_slots, n_opt, device):
    key = (int(n_edges), int(n_patch_slots), int(n_opt), str(device))
    ws = _ws_cache.get(key)
    if ws is None:
        _ws_cache.clear()                                   # one live workspace: the graph size changes once per frame
        ws = _ws_cache[key] = _ba.workspace(*key[:3], device)
    return ws


def _devkey(device):
    d = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def last_status(device=None, clear=True):
    """Status of the most recent BA() on `device` (one host read): 0 = ok, k > 0 = Cholesky breakdown in Gauss-Newton iteration
    k (that iteration and the following ones left poses / patches untouched), -1 = workspace mismatch."""
    dev = _devkey(device)
    if clear:
        _pending[dev] = False
    h = _host.get(dev)
    if h is None:
        return 0
    h[1].synchronize()        # waits for THAT BA's status copy only (long done when the next frame asks), not for the stream
    return int(h[0][0])


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2, check="lazy"):
    """Gauss-Newton bundle adjustment, IN PLACE on the storage of `poses` (an SE3 object or a tensor) and `patches`;
    returns [] like the extension (callers rely on the mutation, devo/devo.py:337).
    A failed factorisation is reported like the reference reports it, by an exception the caller's try / except sees
    (devo/devo.py:336-340) — `check="lazy"` (default): at the NEXT BA() call on the device (after that call's own work has
    been enqueued), so that no call waits for the GPU; `check="now"`: before returning (one host synchronisation); `check="never"`: only through last_status()."""
    pose_data = getattr(poses, "data", poses)
    P = patches.shape[-1]
    n_opt = int(t1) - int(t0)
    if n_opt > MAX_OPTIMISED_POSES:
        raise RuntimeError(f"fastba.BA: {n_opt} optimised poses (t0 = {int(t0)}, t1 = {int(t1)}), at most {MAX_OPTIMISED_POSES} are supported "
                           f"(OPTIMIZATION_WINDOW is 10 in config/default.yaml); shrink the window or use devo_amd.ba.BA")
    dev = _devkey(pose_data.device)
    # lazy check: the status of the PREVIOUS call is read now (its copy is long done) but reported only after THIS call's work is
    # enqueued — a caller that wraps BA() in the reference's try / except (devo.py:336-340) must not lose a healthy adjustment to
    # its predecessor's failure
    prev_code = last_status(pose_data.device) if (check == "lazy" and _pending.get(dev)) else 0
    st = _status.get(dev)
    if st is None:
        st = _status[dev] = torch.zeros(1, dtype=torch.int32, device=pose_data.device)
    ws = _workspace(ii.numel(), patches.numel() // (3 * P * P), n_opt, pose_data.device)
    out = _ba.forward(pose_data, patches, intrinsics, target, weight, lmbda, ii, jj, kk, int(t0), int(t1), int(iterations), ws=ws, status=st)
    h = _host.get(dev)
    if h is None:
        h = _host[dev] = (torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event())
    if not torch.cuda.is_current_stream_capturing():
        h[0].copy_(st, non_blocking=True)
        h[1].record()
    _pending[dev] = check != "never" and not torch.cuda.is_current_stream_capturing()
    if check == "now":
        code = last_status(pose_data.device)
        if code != 0:
            raise BAFailure(f"fastba.BA: bundle adjustment failed (status {code})")
    if prev_code != 0:
        raise BAFailure(f"fastba.BA: the previous bundle adjustment on {dev} failed (status {prev_code}); this call's adjustment has been enqueued")
    return out


def neighbors(ii, jj):
    """[ix, jx]: previous / next edge of the same patch in frame order, -1 where there is none (ba.cpp:104-149)."""
    return _ba.neighbors(ii, jj)


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """coords [1, E, 2, P, P] of every patch pixel in its target frame (ba_cuda.cu:368-418)."""
    return _ba.reproject(poses, patches, intrinsics, ii, jj, kk)
