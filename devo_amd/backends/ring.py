"""Per-slot maintenance of the converted ring buffers (round 6).

The reference keeps its feature pyramid and patch features in ring buffers of `mem` = 32 frames and rewrites ONE slot per frame
(devo/devo.py:523-527: `self.gmap_[self.n % self.mem] = gmap`, `self.fmap1_[:, self.n % self.mem] = ...`; keyframe removal moves a few
slots, :288-291), then looks the whole ring up.  The lookup kernel reads a converted copy of those tensors (channel-blocked fp16 /
split-blocked fp32 levels, the transposed patch operand: DESIGN.md §2), cached per tensor version — so far the WHOLE ring was converted
again for every new frame.

`track_ring_writes()` (called by devo_amd.backends.install()) wraps `torch.Tensor.__setitem__`: a write into a tensor the compiled
binding holds a converted copy of is recorded there as (version counter after the write, element range).  The next lookup converts just
the slots the recorded writes touched — provided EVERY version between the cached one and the current one is accounted for by a record
with a known contiguous range (a `__setitem__` bumps the counter by exactly one).  Anything else (another in-place op, an index the
wrapper cannot turn into one contiguous range, the wrapper not installed) leaves a gap and the whole tensor is converted as before:
the shortcut can be missed, never be wrong.  The wrapper costs one dict-free check per `__setitem__` on CUDA tensors of >= 4 dimensions.

DEVO_RING_SLOTS=0 switches the per-slot path off (both the wrapper and the binding's use of its records)."""
import os
import torch

_orig_setitem = None


def _setitem(self, idx, value):
    r = _orig_setitem(self, idx, value)
    try:
        if self.is_cuda and self.dim() >= 4:
            from . import native
            N = native()
            if N is not None:
                tracked, note = N.cuda_corr._is_tracked, N.cuda_corr._note_write
            else:                                                        # (the ctypes binding keeps its own records: backends/cuda_corr.py)
                from . import cuda_corr as _cc
                tracked, note = _cc.is_tracked, _cc.note_write
            if True:
                p = self.data_ptr()
                if tracked(p):
                    off, length = 0, -1                                  # unknown region unless the index selects ONE contiguous run of this tensor
                    if self.is_contiguous():
                        sub = self[idx]
                        if (sub.is_contiguous() and sub.numel() > 0 and sub.untyped_storage().data_ptr() == self.untyped_storage().data_ptr()
                                and sub.data_ptr() >= p):
                            off, length = (sub.data_ptr() - p) // self.element_size(), sub.numel()
                    note(p, self._version, off, length)
    except Exception:                                                    # noqa: BLE001 — a missing record is a gap: the binding converts everything
        pass
    return r


def track_ring_writes(enable=True):
    """Install (or remove) the `__setitem__` wrapper.  Idempotent; returns whether it is installed afterwards."""
    global _orig_setitem
    if enable and os.environ.get("DEVO_RING_SLOTS", "1") == "0":
        enable = False
    if enable and _orig_setitem is None:
        _orig_setitem = torch.Tensor.__setitem__
        torch.Tensor.__setitem__ = _setitem
    elif not enable and _orig_setitem is not None:
        torch.Tensor.__setitem__ = _orig_setitem
        _orig_setitem = None
    return _orig_setitem is not None


def tracking():
    return _orig_setitem is not None
