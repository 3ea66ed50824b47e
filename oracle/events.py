"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of the step in front of the hot path — SURVEY.md §8(f) row f4:

    utils/event_utils.py:180-232   to_voxel_grid: every event votes into the 2x2x2 neighbouring voxels of (t, y, x) with
                                   weight  polarity * (1-|dx|) * (1-|dy|) * (1-|dt|),  t rescaled to [0, bins-1] in fp64
    utils/voxel_utils.py:6-28      std (== devo/devo.py:438-452): standardise the NON-ZERO voxels of a sequence
                                   (mean / stddev over the non-zeros, fp32 sums), zeros stay zero

Pinned by tests/golden/events_f32.npz (tools/gen_golden_events.py runs the reference's own functions).
"""
import numpy as np
import torch


def to_voxel_grid(xs, ys, ts, ps, H, W, bins=5):
    """xs, ys: pixel coordinates (any float), ts: float64 timestamps (ascending), ps: polarity 0/1 (0 -> -1)."""
    grid = torch.zeros(bins * H * W, dtype=torch.float32)
    # the reference stacks x, y (cast to fp32) with the fp64 timestamps into ONE fp64 array (event_utils.py:201):
    # fp32 values, but every later product runs in fp64
    x = torch.as_tensor(np.asarray(xs, dtype=np.float32).astype(np.float64))
    y = torch.as_tensor(np.asarray(ys, dtype=np.float32).astype(np.float64))
    tt = torch.as_tensor(np.asarray(ts, dtype=np.float64))
    pol = torch.as_tensor(np.asarray(ps).astype(np.int8)).clone()
    pol[pol == 0] = -1
    pol = pol.float()
    t = (tt - tt[0]) * (bins - 1) / (tt[-1] - tt[0])
    for lx in (x.floor(), x.floor() + 1):
        for ly in (y.floor(), y.floor() + 1):
            for lt in (t.floor(), t.floor() + 1):
                ok = (0 <= lx) & (0 <= ly) & (0 <= lt) & (lx <= W - 1) & (ly <= H - 1) & (lt <= bins - 1)
                idx = lx.long() + ly.long() * W + lt.long() * W * H
                w = pol * (1 - (lx - x).abs()) * (1 - (ly - y).abs()) * (1 - (lt - t).abs())     # fp64 product
                grid.index_add_(0, idx[ok], w[ok].float())
    return grid.view(bins, H, W)


def std(voxs, sequence=True):
    """voxs [b, n, c, h, w] -> standardised over the non-zero entries of every sequence (or every frame)."""
    b, n, c, h, w = voxs.shape
    flat = voxs.reshape(b, -1) if sequence else voxs.reshape(b, n, -1)
    nz = flat != 0.0
    cnt = nz.sum(dim=-1)
    if bool(torch.all(cnt > 0)):
        mean = torch.sum(flat, dim=-1, dtype=torch.float32) / cnt
        sd = torch.sqrt(torch.sum(flat ** 2, dim=-1, dtype=torch.float32) / cnt - mean ** 2)
        flat = nz.type_as(flat) * (flat - mean[..., None]) / sd[..., None]
    return flat.reshape(b, n, c, h, w)
