#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator; build container only — it imports the reference).

tests/golden/nms_select.npz: the reference's PatchSelector in its "nms" mode (devo/selector.py:194-287: pooled maxima, boxes, the quadrant
categories, per-frame top m) run on CPU.  The method hard-codes device="cuda" in three tensor constructors and calls
torchvision.ops.batched_nms, which this image does not have: for the run, torch.as_tensor / arange / empty map "cuda" to "cpu", and
batched_nms is the plain greedy loop below (one box at a time, in decreasing score order inside every category — written here,
independently of devo_amd.patchifier.batched_nms, which the fixture then pins together with the logic around it)."""
import os
import sys
import types
import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden_update as GU                            # noqa: E402


def greedy_batched_nms(boxes, scores, idxs, thr):
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    kept = []
    for i in order:
        ok = True
        for j in kept:
            if int(idxs[i]) != int(idxs[j]):
                continue
            a, b = boxes[i], boxes[j]
            iw = max(0.0, min(float(a[2]), float(b[2])) - max(float(a[0]), float(b[0])))
            ih = max(0.0, min(float(a[3]), float(b[3])) - max(float(a[1]), float(b[1])))
            inter = iw * ih
            union = float((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1])) - inter
            if inter / union > thr:
                ok = False
                break
        if ok:
            kept.append(i)
    return torch.tensor(kept, dtype=torch.long)


def main():
    GU.install_shims()
    sys.modules["torchvision.ops"].batched_nms = greedy_batched_nms
    real = {k: getattr(torch, k) for k in ("as_tensor", "arange", "empty")}

    def on_cpu(fn):
        def f(*a, **kw):
            if kw.get("device") == "cuda":
                kw["device"] = "cpu"
            return fn(*a, **kw)
        return f
    for k, fn in real.items():
        setattr(torch, k, on_cpu(fn))
    try:
        from devo.selector import PatchSelector
        g = torch.Generator().manual_seed(11)
        out = {}
        for tag, (n, h, w, m) in {"a": (2, 22, 30, 8), "b": (3, 16, 24, 12)}.items():
            sm = torch.rand(1, n, h, w, generator=g)
            sm[:, :, : h // 3] *= 0.2                                     # structure: weak top rows, a few strong blobs
            sm[:, :, h // 2, w // 2] = 2.0
            out[f"{tag}/scores"] = sm.numpy()
            for grid in (True, False):
                x, y = PatchSelector("nms", grid=grid)(sm, m)
                out[f"{tag}/x_grid{int(grid)}"] = x.long().numpy()
                out[f"{tag}/y_grid{int(grid)}"] = y.long().numpy()
                out[f"{tag}/m"] = m
    finally:
        for k, fn in real.items():
            setattr(torch, k, fn)
    path = os.path.join(ROOT, "tests", "golden", "nms_select.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes", {k: v.shape for k, v in out.items() if hasattr(v, "shape") and "x_" in k})


if __name__ == "__main__":
    main()
