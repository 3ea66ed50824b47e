#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator; build container only — it imports the reference).

Generate tests/golden/events_f32.npz by running the REAL reference functions utils/event_utils.py:to_voxel_grid and
utils/voxel_utils.py:std from /root/reference on CPU (seeded synthetic event streams).  Build container only; the file
holds data (inputs + expected outputs).  Import-time stubs: h5py, numba (jit = identity), torchvision.transforms.functional."""
import os
import sys
import types
import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def main():
    _mod("h5py", File=object)
    _mod("numba", jit=lambda *a, **k: (lambda f: f))
    tv = _mod("torchvision"); tr = _mod("torchvision.transforms"); tv.transforms = tr
    tr.functional = _mod("torchvision.transforms.functional")
    sys.path.insert(0, REF)
    from utils.event_utils import to_voxel_grid
    from utils.voxel_utils import std
    rng = np.random.default_rng(1234)
    out = {}
    for tag, (N, H, W, frac) in {"int": (20000, 48, 64, False), "frac": (15000, 40, 56, True)}.items():
        xs = rng.integers(0, W, N).astype(np.float32)
        ys = rng.integers(0, H, N).astype(np.float32)
        if frac:                                    # rectified coordinates: fractional, some outside the image
            xs = (xs + rng.uniform(-1.5, 1.5, N)).astype(np.float32)
            ys = (ys + rng.uniform(-1.5, 1.5, N)).astype(np.float32)
        ts = np.sort(rng.uniform(1e6, 1.05e6, N)).astype(np.float64)
        ps = rng.integers(0, 2, N).astype(np.int8)
        if frac:
            # the reference indexes with x,y only through floor(): pass floats directly (remapping_maps=None)
            vox = to_voxel_grid(xs, ys, ts, ps.copy(), H=H, W=W, nb_of_time_bins=5)
        else:
            vox = to_voxel_grid(xs.astype(np.int64), ys.astype(np.int64), ts, ps.copy(), H=H, W=W, nb_of_time_bins=5)
        for k, v in dict(xs=xs, ys=ys, ts=ts, ps=ps, vox=vox.numpy()).items():
            out[f"{tag}/{k}"] = v
        seq = torch.stack([vox, vox.flip(0) * 0.5])[None]                       # [1, 2, 5, H, W]
        out[f"{tag}/std_seq"] = std(seq.clone(), sequence=True).numpy()
        out[f"{tag}/std_frame"] = std(seq.clone(), sequence=False).numpy()
    path = os.path.join(ROOT, "tests", "golden", "events_f32.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
