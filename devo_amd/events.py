"""Event stream -> voxel grid and voxel standardisation on the GPU, with the reference's function
names and argument meaning: utils/event_utils.py:180-232 `to_voxel_grid`, utils/voxel_utils.py:6-28 `std`
(== the NORM='std' branch of devo/devo.py:438-452).  Inputs are device tensors; no CPU fallback."""
import torch
from . import _lib as L


def to_voxel_grid(xs, ys, ts, ps, H=480, W=640, nb_of_time_bins=5):
    """xs, ys: pixel coordinates [N] (any numeric dtype; rectified = fractional allowed), ts: timestamps [N] ascending,
    ps: polarity [N] (0 / 1 or -1 / 1) -> voxel grid [nb_of_time_bins, H, W] float32 on the same GPU."""
    L.require_gpu(xs, ys, ts, ps)
    N = xs.numel()
    x = xs.reshape(-1).float().contiguous()
    y = ys.reshape(-1).float().contiguous()
    t = ts.reshape(-1).double().contiguous()
    p = ps.reshape(-1).to(torch.int8).contiguous()
    grid = torch.empty(int(nb_of_time_bins), int(H), int(W), dtype=torch.float32, device=xs.device)
    rc = L.lib().devo_voxelize(L.ptr(x), L.ptr(y), L.ptr(t), L.ptr(p), N, int(H), int(W), int(nb_of_time_bins), L.ptr(grid), L.stream())
    L.check(rc, "events.to_voxel_grid")
    return grid


def std(voxs, sequence=True):
    """voxs [b, n, c, h, w] float32 -> standardised copy (non-zero voxels of every sequence, or of every frame)."""
    L.require_gpu(voxs)
    b, n, c, h, w = voxs.shape
    out = voxs.float().contiguous().clone()
    nseg = b if sequence else b * n
    ws = torch.empty(L.lib().devo_voxel_std_workspace_bytes(nseg), dtype=torch.uint8, device=voxs.device)
    rc = L.lib().devo_voxel_std(L.ptr(out), nseg, out.numel() // max(nseg, 1), L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, "events.std")
    return out.view(b, n, c, h, w)
