"""`devo.fastba` as the rest of DEVO imports it (devo/devo.py:13, devo/enet.py:11): three callables over the HIP bundle
adjustment.  The reference module binds them straight to its CUDA extension (devo/fastba/ba.py:4-8); here they go
through devo_amd.backends.cuda_ba and keep one BA workspace alive between calls of the same size, so that the update
loop does not allocate on every iteration."""
import torch
from .backends import cuda_ba as _ba

_ws_cache = {}
_status = {}                  # device -> int32 [1]: 0 = ok, k > 0 = the Cholesky factorisation broke down in iteration k, -1 = bad workspace
_host = {}                    # device -> two (pinned int32 [1], event) slots: the status is copied out asynchronously after every BA()
_pending = {}                 # device -> True while the status of the last BA() has not been looked at
import os as _os
PINNED_STATUS = _os.environ.get("DEVO_BA_PINNED_STATUS", "1") != "0"   # 0: a device status word + an asynchronous copy behind every BA() (rounds 1-5)
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


def _slots(dev):
    h = _host.get(dev)
    if h is None:                                            # two (pinned int32 [1], event, [in flight]) slots, used alternately
        h = _host[dev] = {"slots": [[torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event(), False] for _ in range(2)], "next": 0}
    return h


def _collect(dev, wait_all):
    """First non-zero status among the finished status copies on `dev` (their slots become free).  wait_all=False: the OLDER slot is waited
    for when both are in flight (it belongs to the call before the previous one: done unless the host runs two adjustments ahead of the GPU),
    the newer one is only polled — so that a loop of BA() calls never waits for the adjustment it has just enqueued."""
    h = _host.get(dev)
    if h is None:
        return 0
    code = 0
    order = [h["slots"][(h["next"] + k) % 2] for k in range(2)]              # oldest first
    busy = [sl for sl in order if sl[2]]
    for k, sl in enumerate(busy):
        must = wait_all or (len(busy) == 2 and k == 0)
        if must:
            sl[1].synchronize()
        elif not sl[1].query():
            continue
        sl[2] = False
        if code == 0:
            code = int(sl[0][0])
    return code


def last_status(device=None, clear=True):
    """Status of the BA() calls on `device` whose status has not been looked at yet (waits for their status copies, not for the stream): 0 = ok,
    k > 0 = Cholesky breakdown in Gauss-Newton iteration k (that iteration and the following ones left poses / patches untouched), -1 =
    workspace mismatch."""
    dev = _devkey(device)
    if clear:
        _pending[dev] = False
    return _collect(dev, wait_all=True)


def BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations=2, check="lazy"):
    """Gauss-Newton bundle adjustment, IN PLACE on the storage of `poses` (an SE3 object or a tensor) and `patches`;
    returns [] like the extension (callers rely on the mutation, devo/devo.py:337).
    A failed factorisation is reported like the reference reports it, by an exception the caller's try / except sees
    (devo/devo.py:336-340) — `check="lazy"` (default): by a LATER BA() call on the device, after that call's own work has been enqueued and
    without waiting for the GPU: the next call if the failed adjustment has finished by then, else the one after it (an update loop runs
    ahead of the GPU; waiting for the adjustment just enqueued would serialise host and device — 257 against 2xx us per iteration of the
    reference's sequence); `check="now"`: before returning (one host synchronisation); `check="never"`: only through last_status()."""
    pose_data = getattr(poses, "data", poses)
    P = patches.shape[-1]
    n_opt = int(t1) - int(t0)
    if n_opt > MAX_OPTIMISED_POSES:
        raise RuntimeError(f"fastba.BA: {n_opt} optimised poses (t0 = {int(t0)}, t1 = {int(t1)}), at most {MAX_OPTIMISED_POSES} are supported "
                           f"(OPTIMIZATION_WINDOW is 10 in config/default.yaml); shrink the window or use devo_amd.ba.BA")
    dev = _devkey(pose_data.device)
    # lazy check: finished status copies of EARLIER calls are read now but reported only after THIS call's work is enqueued — a caller
    # that wraps BA() in the reference's try / except (devo.py:336-340) must not lose a healthy adjustment to a predecessor's failure
    prev_code = _collect(dev, wait_all=False) if (check == "lazy" and _pending.get(dev)) else 0
    ws = _workspace(ii.numel(), patches.numel() // (3 * P * P), n_opt, pose_data.device)
    capturing = torch.cuda.is_current_stream_capturing()
    if not capturing and PINNED_STATUS:
        # the kernels write the status word straight into one of the two pinned host words (device-visible at its own address): no fill in
        # front of the call (its first solver launch resets the word), no copy behind it — two launches of ~5 us of GPU time less per adjustment
        h = _slots(dev)
        sl = h["slots"][h["next"]]
        if sl[2]:                                            # (check="never" loops: the slot's old status is dropped with its wait)
            sl[1].synchronize()
        out = _ba.forward(pose_data, patches, intrinsics, target, weight, lmbda, ii, jj, kk, int(t0), int(t1), int(iterations), ws=ws, status=sl[0])
        sl[1].record()
        sl[2] = True
        h["next"] = (h["next"] + 1) % 2
    else:
        st = _status.get(dev)
        if st is None:
            st = _status[dev] = torch.zeros(1, dtype=torch.int32, device=pose_data.device)
        out = _ba.forward(pose_data, patches, intrinsics, target, weight, lmbda, ii, jj, kk, int(t0), int(t1), int(iterations), ws=ws, status=st)
        if not capturing:
            h = _slots(dev)
            sl = h["slots"][h["next"]]
            if sl[2]:
                sl[1].synchronize()
            sl[0].copy_(st, non_blocking=True)
            sl[1].record()
            sl[2] = True
            h["next"] = (h["next"] + 1) % 2
    _pending[dev] = check != "never" and not capturing
    if check == "now":
        code = last_status(pose_data.device)
        if code != 0:
            raise BAFailure(f"fastba.BA: bundle adjustment failed (status {code})")
    if prev_code != 0:
        raise BAFailure(f"fastba.BA: an earlier bundle adjustment on {dev} failed (status {prev_code}); this call's adjustment has been enqueued")
    return out


def neighbors(ii, jj):
    """[ix, jx]: previous / next edge of the same patch in frame order, -1 where there is none (ba.cpp:104-149)."""
    return _ba.neighbors(ii, jj)


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """coords [1, E, 2, P, P] of every patch pixel in its target frame (ba_cuda.cu:368-418)."""
    return _ba.reproject(poses, patches, intrinsics, ii, jj, kk)
