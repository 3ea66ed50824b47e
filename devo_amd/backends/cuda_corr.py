"""HIP-backed module with the interface of the reference's `cuda_corr` extension
(devo/altcorr/correlation.cpp:57-63): forward, backward, patchify_forward, patchify_backward.
Every function allocates its outputs with torch and enqueues ONE fused kernel of libdevo_hip.so on the
current stream.  No CPU fallback."""
import ctypes
import torch
from .. import _lib as L


def _prep(fmap1, fmap2, coords, ii, jj, allow_blocked=False):
    L.require_gpu(fmap1, fmap2, coords, ii, jj)
    if fmap1.dtype != fmap2.dtype:
        raise RuntimeError("cuda_corr: fmap1 and fmap2 must have the same dtype")
    if fmap1.dim() != 5 or fmap2.dim() not in ((5, 6) if allow_blocked else (5,)) or coords.dim() != 5:
        raise RuntimeError("cuda_corr: expected fmap1 [B,Np,C,P,P], fmap2 [B,n,C,H,W], coords [B,E,2,P,P]")
    fmap1 = fmap1.contiguous()
    coords = coords.float().contiguous()
    ii = ii.long().contiguous()
    jj = jj.long().contiguous()
    return fmap1, fmap2, coords, ii, jj


import os
# dense-product lookup kernel (corr_mm.h); False: the 4x4 matrix-core kernel (DEVO_CORR_MFMA=0: neither — the staged kernel; both are read
# by the library as well)
MM_KERNEL = os.environ.get("DEVO_CORR_MM", "1") != "0" and os.environ.get("DEVO_CORR_MFMA", "1") != "0"
PLAN_MIN_EDGES = 2048      # below this the lookup is launch-bound and a locality plan cannot pay for itself
_last_plan = None           # (key, plan) of the last per-level call that made a plan
NCHW_CONVERT_MIN_EDGES = 1024    # from here on a lookup into the reference's NCHW pyramid goes through a cached channel-blocked copy

_blocked_cache = {}        # (ptr, version, shape, strides, dtype) -> (source tensor [kept alive], channel-blocked copy); LRU
BLOCKED_CACHE_ENTRIES = 4

# ---- per-slot maintenance of a converted ring in THIS binding (round 6; csrc/bind.cpp has the same for the compiled one: DESIGN 2).  A write
# through the `Tensor.__setitem__` wrapper of backends/ring.py into a tensor this cache holds a converted copy of is recorded as (version counter
# after the write, element range); a lookup whose key differs from a cached entry only in the version converts just the frames / patches the
# records cover — provided EVERY version in between has a record with a known contiguous range.  Anything else: the whole tensor, as before.
import os as _os
RING_SLOTS = _os.environ.get("DEVO_RING_SLOTS", "1") != "0"
MAX_WRITE_RECS = 512
_writes = {}               # data_ptr -> [(version after the write, first element, elements | -1 = unknown)]
_conv_stats = [0, 0, 0, 0, 0]     # whole levels, frames of levels, whole patch operands, patch ranges, patches in those ranges


def is_tracked(ptr):
    return any(k[0] == ptr for k in _blocked_cache) or any(k[0] == ptr for k in _patch_t_cache)


def note_write(ptr, version_after, off, length):
    recs = _writes.setdefault(ptr, [])
    if len(recs) >= MAX_WRITE_RECS:
        recs.clear()                                          # (a gap: the next lookup converts everything)
    recs.append((int(version_after), int(off), int(length)))


def _forget_writes(ptr, upto):
    recs = _writes.get(ptr)
    if recs is not None:
        recs[:] = [r for r in recs if r[0] > upto]
        if not recs:
            del _writes[ptr]


def _writes_between(ptr, v0, v1):
    """every version in (v0, v1] has a record with a known range -> their (first, end) element ranges; else None"""
    need = v1 - v0
    recs = _writes.get(ptr)
    if need <= 0 or need > MAX_WRITE_RECS or not recs:
        return None
    seen, ranges = set(), []
    for ver, off, length in recs:
        if v0 < ver <= v1:
            if length < 0 or ver in seen:
                return None
            seen.add(ver)
            ranges.append((off, off + length))
    return ranges if len(seen) == need else None


def _slot_path_ok(t):
    return RING_SLOTS and t.is_contiguous() and L.lib().devo_stream_capturing(L.stream()) == 0


def convert_stats():
    """(whole levels, frames of levels, whole patch operands, patch ranges, patches in those ranges) converted so far by the active binding"""
    N = _nat()
    return tuple(N.cuda_corr._convert_stats()) if N is not None else tuple(_conv_stats)


def clear_caches():
    """Drop the converted pyramid copies and patch operands of the active binding (tests)."""
    N = _nat()
    if N is not None:
        N.clear_caches()
        return
    _blocked_cache.clear()
    _patch_t_cache.clear()
    _writes.clear()
_MM_DEFAULT = MM_KERNEL


def _nat():
    """The compiled binding (devo_amd._C) when it is present and this module's switches are at their defaults (MM_KERNEL is read once,
    from the environment, by the compiled binding)."""
    from . import native
    return native() if MM_KERNEL == _MM_DEFAULT else None


class SplitLevel:
    """An fp32 pyramid level in the split-blocked format of devo_corr_pyramid_split: `data` float32-sized [B, n, C/8, H, W, 8] whose
    32-byte pixel blocks hold fp16 (hi0..7 | lo0..7), `exps` int32 [B n (+ n scratch)] = one scale exponent per frame.  What the
    dense-product lookup kernel multiplies; an opaque operand like patches_transposed()'s."""
    __slots__ = ("data", "exps", "shape", "dtype", "is_cuda", "device")

    def __init__(self, data, exps, C):
        self.data, self.exps = data, exps
        B, n, _, H, W, _ = data.shape
        self.shape = (B, n, C, H, W)
        self.dtype, self.is_cuda, self.device = torch.float32, True, data.device

    def dim(self):
        return 5


def split_level(fmap2):
    """fp32 fmap2 [B, n, C, H, W] (any strides) or channel-blocked [B, n, C/cb, H, W, cb] -> SplitLevel (two launches per batch entry)."""
    L.require_gpu(fmap2)
    cblock = 0
    if fmap2.dim() == 6:
        cblock = fmap2.shape[5]
        B, n, nblk, H, W, _ = fmap2.shape
        C = nblk * cblock
        if fmap2.stride(5) != 1:
            raise RuntimeError("cuda_corr.split_level: malformed channel-blocked fmap2")
    else:
        B, n, C, H, W = fmap2.shape
    if fmap2.dtype != torch.float32 or C % 8:
        raise RuntimeError("cuda_corr.split_level: fp32 with C % 8 == 0 only")
    data = torch.empty(B, n, C // 8, H, W, 8, dtype=torch.float32, device=fmap2.device)
    exps = torch.empty(B * n + n, dtype=torch.int32, device=fmap2.device)
    st = fmap2.stride()
    for b in range(B):                                        # ascending: call b's scratch is the (not yet written) slice of b + 1
        rc = L.lib().devo_corr_pyramid_split(L.ptr(fmap2[b]), L.i64arr([st[1], st[2], st[3], st[4]]), int(cblock), n, C, H, W,
                                             L.ptr(data[b]), data.stride(1), ctypes.c_void_p(exps.data_ptr() + 4 * b * n), L.stream())
        L.check(rc, "cuda_corr.split_level")
    return SplitLevel(data, exps, C)


def _mm_wants_split(fmap2, C):
    return MM_KERNEL and fmap2.dtype == torch.float32 and C % 32 == 0 and C <= 128


def _fast_layout(fmap2, n_edges, allow_split=True):
    """The reference keeps its feature pyramid NCHW ([1, 32, 128, H, W] fp16 ring, devo/devo.py:71-83, rewritten in place one
    frame per step); read directly, every channel of a pixel is H*W elements away and the lookup falls back to the strided
    generic kernel — measured 24x slower than the matrix-core kernel on the channel-blocked layout (profiles/README.md, r02x).
    So an NCHW level is converted ONCE per version of the tensor into a channel-blocked copy [B, n, C/8, H, W, 8]
    (devo_pyramid_build: one pass, ~60 us for DEVO's whole fp16 ring) and cached: the key is (storage address, version counter,
    shape, strides, dtype) and the cache keeps the source tensor alive, so an equal key is the same storage with the same
    contents (in-place writes such as `fmap1_[:, k] = f` bump the version counter).  fp16 channels-last / channel-blocked inputs,
    small edge lists, other dtypes: returned unchanged.  DEVO_CORR_NCHW_DIRECT=1 disables the conversion.
    fp32 levels of ANY layout (NCHW, channels-last, channel-blocked) that the dense-product kernel can take (C % 32 == 0, C <= 128) are
    converted — same cache, same key — into the split-blocked format (SplitLevel: fp16 hi | lo of the value scaled per frame): the
    kernel then loads its operands instead of splitting every value it reads, at any magnitude of the features."""
    import os
    if isinstance(fmap2, SplitLevel):
        return fmap2
    N = _nat()
    if N is not None and fmap2.is_cuda:                       # ONE cache for both bindings: the compiled one's
        data, exps, cblock = N.cuda_corr._fast_layout(fmap2, int(n_edges), bool(allow_split))
        if cblock == L.CBLOCK_SPLIT8:
            return SplitLevel(data, exps, data.shape[2] * 8)
        return data
    if fmap2.dtype not in (torch.float16, torch.float32) or n_edges <= 0:
        return fmap2
    st = fmap2.stride()
    C = fmap2.shape[2] * (fmap2.shape[5] if fmap2.dim() == 6 else 1)
    want_split = allow_split and _mm_wants_split(fmap2, C) and fmap2.numel() > 0
    if not want_split:
        if fmap2.dim() != 5 or n_edges < NCHW_CONVERT_MIN_EDGES or os.environ.get("DEVO_CORR_NCHW_DIRECT", "0") == "1":
            return fmap2
        B, n, C, H, W = fmap2.shape
        if C % 8 or tuple(st[2:]) != (H * W, W, 1) or B * n == 0:
            return fmap2                                      # not plain NCHW frames (e.g. channels-last already)
    key = (fmap2.data_ptr(), fmap2._version, tuple(fmap2.shape), tuple(st), fmap2.dtype)
    hit = _blocked_cache.pop(key, None)
    if hit is not None:
        _blocked_cache[key] = hit                             # (most recently used last)
        return hit[1]
    older = [k for k in _blocked_cache if k[0] == key[0]]
    if fmap2.dim() == 5 and _slot_path_ok(fmap2):             # the same ring at an older version whose every write since is on record
        for k in older:
            if k[2:] != key[2:]:
                continue
            ranges = _writes_between(key[0], k[1], key[1])
            if ranges is None:
                break
            _, conv = _blocked_cache.pop(k)
            Bq, nq, Cq, Hq, Wq = fmap2.shape
            per, total = Cq * Hq * Wq, Bq * nq
            dirty = [False] * total
            for lo, hi in ranges:
                f = max(0, lo // per)
                while f < total and f * per < hi:
                    dirty[f] = True
                    f += 1
            f = 0
            while f < total:
                if not dirty[f]:
                    f += 1
                    continue
                f1 = f
                while f1 + 1 < total and dirty[f1 + 1] and (f1 + 1) // nq == f // nq:
                    f1 += 1
                b, f0, cnt = f // nq, f % nq, f1 - f + 1
                if want_split:
                    scratch = torch.empty(cnt, dtype=torch.int32, device=fmap2.device)
                    rc = L.lib().devo_corr_pyramid_split_frames(L.ptr(fmap2[b, f0]), L.i64arr([st[1], st[2], st[3], st[4]]), 0, cnt, Cq, Hq, Wq,
                                                                L.ptr(conv.data[b, f0]), conv.data.stride(1),
                                                                ctypes.c_void_p(conv.exps.data_ptr() + 4 * (b * nq + f0)), L.ptr(scratch), L.stream())
                    L.check(rc, "cuda_corr: fp32 ring slot -> split-blocked")
                else:
                    rc = L.lib().devo_pyramid_build(L.ptr(fmap2[b, f0]), L.ptr(conv[b, f0]), None, cnt, Cq, Hq, Wq, st[1], conv.stride(1), 0,
                                                    L.dtype_code(fmap2), L.stream())
                    L.check(rc, "cuda_corr: NCHW ring slot -> channel-blocked")
                _conv_stats[1] += cnt
                f = f1 + 1
            for k2 in [k2 for k2 in _blocked_cache if k2[0] == key[0]]:
                del _blocked_cache[k2]
            _blocked_cache[key] = (fmap2, conv)               # (the same converted tensors under the new version)
            _forget_writes(key[0], key[1])
            return conv
    for k in older:
        _blocked_cache.pop(k, None)                           # an older version of this tensor
    while len(_blocked_cache) >= BLOCKED_CACHE_ENTRIES:       # least recently used first (dicts keep insertion order): a DEVO
        del _blocked_cache[next(iter(_blocked_cache))]        # process holds two levels of one ring = 2 entries, ~300 MB in fp16
    _conv_stats[0] += 1
    _forget_writes(key[0], key[1])
    if want_split:
        lvl = split_level(fmap2)
        _blocked_cache[key] = (fmap2, lvl)
        return lvl
    blk = torch.empty(B, n, C // 8, H, W, 8, dtype=fmap2.dtype, device=fmap2.device)
    for b in range(B):
        rc = L.lib().devo_pyramid_build(L.ptr(fmap2[b]), L.ptr(blk[b]), None, n, C, H, W, st[1], blk.stride(1), 0,
                                        L.dtype_code(fmap2), L.stream())
        L.check(rc, "cuda_corr: NCHW -> channel-blocked")
    _blocked_cache[key] = (fmap2, blk)
    return blk


def cached_levels():
    """Pyramid levels held in converted form (whichever binding is active keeps the cache): for tests."""
    N = _nat()
    return N.cuda_corr._cached_levels() if N is not None else len(_blocked_cache)


def plan_buffer(n_slots, device, groups=False):
    """The plan buffer of devo_corr_order: int32 [2 n + 2] = n edge slots | number of heavy slots (in front) | n ints of scratch |
    number of dead slots (at the end; group plans); a GROUP plan has PLAN_TAIL more ints behind them (the groups' first slots)."""
    return torch.empty(2 * int(n_slots) + 2 + (L.PLAN_TAIL if groups else 0), dtype=torch.int32, device=device)


def plan_kind(order, n_slots):
    """DEVO_PLAN_GROUPS for a buffer with a group plan's tail (plan(..., l1 = 4)), else DEVO_PLAN_EDGES."""
    return L.PLAN_GROUPS if (order is not None and order.numel() == 2 * int(n_slots) + 2 + L.PLAN_TAIL) else L.PLAN_EDGES


def plan(coords, jj, n_frames, height, coord_scale=1.0, radius=3, width=0, l1=0):
    """Locality plan (devo_corr_order): edge slots sorted by (target frame, 16-row band, 8-px column).  One plan serves every
    level of a pyramid; `coords / coord_scale` must be the coordinates of the level with `height` rows.
    width, l1: GROUP plan for forward_pyramid — the level is `width` wide and the lookup has a second level at 1 / l1 of its
    resolution (DEVO: 4), radius 3: edges sorted by (target frame, tile of 6 x 6 level-1 cells of the patch centre) with every group's
    first slot behind them; the fused lookup then reads level 1 from LDS regions shared by a group's edges.  Geometries without a
    group plan (too many groups, other radii) get an edge plan."""
    L.require_gpu(coords, jj)
    coords = coords.float().contiguous()
    jj = jj.long().contiguous()
    B, E = coords.shape[:2]
    groups = int(l1) >= 2 and group_plan_supported(B, n_frames, height, width, l1, radius)
    order = plan_buffer(B * E, coords.device, groups)
    rc = L.lib().devo_corr_order(L.ptr(coords), L.ptr(jj), L.ptr(order), B, E, int(n_frames), coords.shape[3], int(height),
                                 float(coord_scale), int(radius), int(width) if groups else 0, int(l1) if groups else 0, L.stream())
    L.check(rc, "cuda_corr.plan")
    return order


GROUP_TILE = 6             # corr_tile.h CORR_GRP_T


def group_plan_supported(batch, n_frames, height, width, l1, radius):
    """corr_tile.h corr_grp_nbins: radius 3, a quarter-resolution second level, at most 4095 groups of 6 x 6 level-1 cells."""
    if int(radius) != 3 or int(l1) != 4 or int(width) <= 0 or os.environ.get("DEVO_CORR_GROUP", "1") == "0":
        return False
    h1, w1 = int(height) // 4, int(width) // 4
    if h1 < 1 or w1 < 1:
        return False
    g = lambda c: (c + GROUP_TILE - 1) // GROUP_TILE
    return int(batch) * int(n_frames) * g(h1) * g(w1) + 1 <= 4096


def plan_finish(order, jj, n_frames, height, radius=3, batch=1, width=0, l1=0):
    """Second half of plan() for a buffer whose bins cuda_ba.transform(..., plan_for=...) has already written (width, l1: as given there)."""
    L.require_gpu(order, jj)
    jj = jj.long().contiguous()
    E = jj.numel()
    groups = int(l1) >= 2 and order.numel() == 2 * batch * E + 2 + L.PLAN_TAIL      # (transform falls back to an edge plan where no group plan exists)
    rc = L.lib().devo_corr_order(None, L.ptr(jj), L.ptr(order), batch, E, int(n_frames), 3, int(height), 1.0, int(radius),
                                 int(width) if groups else 0, int(l1) if groups else 0, L.stream())
    L.check(rc, "cuda_corr.plan_finish")
    return order


_patch_t_cache = {}        # (ptr, version, shape, dtype) -> (source tensor [kept alive], transposed copy [B, Np, 9, C])


def patches_transposed(fmap1):
    """fmap1 [B, Np, C, 3, 3] -> the opaque patch operand of the dense-product lookup kernel (devo_corr_patch_transpose; a uint8 buffer:
    fp16 [B, Np, 9, C], fp32 split records + one scale exponent per patch).  DEVO's patch features change once per frame, not per update iteration: the copy is cached per version of the tensor
    (same key discipline as _fast_layout)."""
    N = _nat()
    if N is not None and fmap1.is_cuda:
        t = N.cuda_corr._patch_operand(fmap1)
        if t is None:
            raise RuntimeError("cuda_corr.patches_transposed: no patch operand for this tensor (C % 32, fp16 / fp32, P = 3)")
        return t
    key = (fmap1.data_ptr(), fmap1._version, tuple(fmap1.shape), fmap1.dtype)
    hit = _patch_t_cache.pop(key, None)
    if hit is not None:
        _patch_t_cache[key] = hit                             # (most recently used last)
        return hit[1]
    older = [k for k in _patch_t_cache if k[0] == key[0]]
    if _slot_path_ok(fmap1):
        for k in older:
            if k[2:] != key[2:]:
                continue
            ranges = _writes_between(key[0], k[1], key[1])
            if ranges is None:
                break
            _, t = _patch_t_cache.pop(k)
            n_p, Cq = fmap1.shape[0] * fmap1.shape[1], fmap1.shape[2]
            per = Cq * 9
            for lo, hi in ranges:
                p0, p1 = max(0, lo // per), min(n_p, (hi + per - 1) // per)
                if p1 <= p0:
                    continue
                rc = L.lib().devo_corr_patch_transpose_range(L.ptr(fmap1), L.ptr(t), n_p, p0, p1 - p0, Cq, L.dtype_code(fmap1), L.stream())
                L.check(rc, "cuda_corr.patches_transposed (slot)")
                _conv_stats[3] += 1
                _conv_stats[4] += p1 - p0
            for k2 in [k2 for k2 in _patch_t_cache if k2[0] == key[0]]:
                del _patch_t_cache[k2]
            _patch_t_cache[key] = (fmap1, t)
            _forget_writes(key[0], key[1])
            return t
    for k in older:
        _patch_t_cache.pop(k, None)                           # an older version of this tensor
    while len(_patch_t_cache) >= BLOCKED_CACHE_ENTRIES:       # least recently used first
        del _patch_t_cache[next(iter(_patch_t_cache))]
    _conv_stats[2] += 1
    _forget_writes(key[0], key[1])
    B, Np, C = fmap1.shape[:3]
    nbytes = int(L.lib().devo_corr_patch_operand_bytes(B * Np, C, L.dtype_code(fmap1)))      # (fp32: split records + one exponent per patch)
    if nbytes == 0 and B * Np > 0:
        raise RuntimeError(f"cuda_corr.patches_transposed: C = {C} unsupported (C % 8)")
    t = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=fmap1.device)
    rc = L.lib().devo_corr_patch_transpose(L.ptr(fmap1), L.ptr(t), B * Np, C, L.dtype_code(fmap1), L.stream())
    L.check(rc, "cuda_corr.patches_transposed")
    _patch_t_cache[key] = (fmap1, t)
    return t


def _patch_operand(fmap1, C, P):
    """The dense-product kernel (corr_mm.h) wants the patches as devo_corr_patch_transpose lays them out ([Np, 9, C]; cached per
    version of fmap1).  None: the lookup takes the 4x4 matrix-core kernel or the staged / generic ones."""
    if MM_KERNEL and P == 3 and fmap1.shape[3] == 3 and C % 32 == 0 and fmap1.dtype in (torch.float16, torch.float32) and fmap1.numel() > 0:
        return patches_transposed(fmap1)
    return None


def forward_into(out, fmap1, fmap2, coords, ii, jj, radius, estride, lstride, offset, order=None, coord_div=1.0):
    """corr forward writing element l of edge (b,e) at out[(b*E+e)*estride + l*lstride + offset].
    coord_div: the kernel looks up at coords / coord_div (same IEEE division as `coords / s` on the tensor, without
    materialising it)."""
    fmap1, fmap2, coords, ii, jj = _prep(fmap1, fmap2, coords, ii, jj, allow_blocked=True)
    B, E = coords.shape[:2]
    P = coords.shape[3]
    _, Np, C = fmap1.shape[:3]
    f1t = _patch_operand(fmap1, C, P)
    # (fp32: the split-blocked copy only when the dense-product kernel will take the call; the exact kernels read the raw level)
    fmap2 = _fast_layout(fmap2, B * E, allow_split=f1t is not None and lstride > 0)
    if order is None and B * E >= PLAN_MIN_EDGES:
        # DEVO calls corr once per pyramid level with the same index tensors (devo.py:215-216): the second call takes the plan the first one
        # made (one plan serves every level; the result does not depend on the plan by one bit) — handed on once, to the call right behind
        global _last_plan
        key = (jj.data_ptr(), jj._version, B * E, fmap2.shape[1], int(radius), str(jj.device), L.stream().value)
        if _last_plan is not None and _last_plan[0] == key:
            order, _last_plan = _last_plan[1], None
        else:
            order = plan(coords, jj, fmap2.shape[1], fmap2.shape[3], float(coord_div), radius)
            _last_plan = (key, order)
    H2, W2, strides, cblock, data, exps = _level_desc(fmap2, C)
    n2 = fmap2.shape[1]
    rc = L.lib().devo_corr_forward(L.ptr(fmap1), L.ptr(data), L.ptr(coords), L.ptr(ii), L.ptr(jj), L.ptr(out),
                                   B, E, Np, n2, C, P, H2, W2, L.i64arr(strides), cblock, estride, lstride, offset,
                                   int(radius), L.dtype_code(fmap1), L.ptr(order), float(coord_div), L.ptr(f1t), L.ptr(exps), L.stream())
    L.check(rc, "cuda_corr.forward")


def forward(fmap1, fmap2, coords, ii, jj, radius):
    """correlation.cpp:58: the compiled binding's function when present (devo_amd._C.cuda_corr.forward), else the ctypes form below."""
    N = _nat()
    if N is not None and PLAN_MIN_EDGES == 2048:
        return N.cuda_corr.forward(fmap1, fmap2, coords, ii, jj, int(radius))
    return _forward_ctypes(fmap1, fmap2, coords, ii, jj, radius)


def _forward_ctypes(fmap1, fmap2, coords, ii, jj, radius):
    """correlation.cpp:58.  Returns [corr] with logical shape [B, E, 2r+1 (x offset), 2r+1 (y offset), P, P]
    (the reference returns the same logical tensor as a permuted view; here it is contiguous)."""
    B, E = coords.shape[:2]
    P = coords.shape[3]
    Dm = 2 * int(radius) + 1
    out = torch.empty(B, E, Dm, Dm, P, P, dtype=fmap1.dtype, device=fmap1.device)
    forward_into(out, fmap1, fmap2, coords, ii, jj, radius, Dm * Dm * P * P, 1, 0)
    return [out]


def _level_desc(fmap2, C):
    """(H, W, strides[5], cblock, storage tensor, exponents or None) of one pyramid level as the C ABI wants them."""
    if isinstance(fmap2, SplitLevel):
        d = fmap2.data
        if d.shape[2] * 8 != C:
            raise RuntimeError("cuda_corr: split-blocked fmap2 does not match fmap1's channels")
        return d.shape[3], d.shape[4], [int(x) for x in d.stride()[:5]], L.CBLOCK_SPLIT8, d, fmap2.exps
    cblock, strides = 0, fmap2.stride()
    if fmap2.dim() == 6:                      # channel-blocked storage [B, n, C/cb, H, W, cb] (altcorr.channel_blocked)
        cblock = fmap2.shape[5]
        if fmap2.stride(5) != 1 or fmap2.shape[2] * cblock != C:
            raise RuntimeError("cuda_corr: malformed channel-blocked fmap2")
        strides = strides[:5]
    return fmap2.shape[3], fmap2.shape[4], [int(x) for x in strides], int(cblock), fmap2, None


def forward_pyramid(fmap1, pyramid, coords, ii, jj, radius, scales, out=None, order=None):
    """Fused equivalent of devo/devo.py:215-217 / enet.py:212-216:
    torch.stack([corr(fmap1, pyr[l], coords / scales[l], ...) for l], -1).view(B, E, -1) without the per-level
    tensors, the scaled coordinate tensors or the stack copy: each level's workgroups write their interleaved slice
    directly.  Two levels in a layout the staged kernel reads go out as ONE launch (devo_corr_forward_pyramid2)."""
    B, E = coords.shape[:2]
    P = coords.shape[3]
    Dm = 2 * int(radius) + 1
    nl = len(pyramid)
    per = Dm * Dm * P * P
    if out is None:
        out = torch.empty(B, E, per * nl, dtype=fmap1.dtype, device=fmap1.device)
    raw = pyramid
    fused = nl == 2 and B * E > 0 and pyramid[0].dtype == pyramid[1].dtype and pyramid[0].dtype in (torch.float32, torch.float16) and fmap1.is_cuda
    f1t = _patch_operand(fmap1.contiguous(), fmap1.shape[2], P) if fused else None
    # (fp32: the split-blocked copies only when the dense-product kernel will take the call; the exact kernels read the raw levels)
    pyramid = [_fast_layout(f, B * E, allow_split=f1t is not None) if f.is_cuda else f for f in pyramid]
    if fused and isinstance(pyramid[0], SplitLevel) != isinstance(pyramid[1], SplitLevel):
        pyramid = [_fast_layout(f, B * E, allow_split=False) for f in raw]
    if order is None and B * E >= PLAN_MIN_EDGES:
        # an edge plan; DEVO_CORR_GROUP=1: a group plan where the group form can run (level 1 from LDS regions: measured slower, opt-in)
        grp = fused and float(scales[0]) == 1.0 and float(scales[1]) == 4.0 and os.environ.get("DEVO_CORR_GROUP", "0") == "1"
        order = plan(coords, jj, pyramid[0].shape[1], pyramid[0].shape[3], scales[0], radius, width=pyramid[0].shape[4] if grp else 0, l1=4 if grp else 0)
    if fused:
        f1, f2a, c_, ii_, jj_ = _prep(fmap1, raw[0], coords, ii, jj, allow_blocked=True)
        _prep(fmap1, raw[1], coords, ii, jj, allow_blocked=True)
        C, Np = f1.shape[2], f1.shape[1]
        d0, d1 = _level_desc(pyramid[0], C), _level_desc(pyramid[1], C)
        hw = (ctypes.c_int * 4)(d0[0], d0[1], d1[0], d1[1])
        cb = (ctypes.c_int * 2)(d0[3], d1[3])
        cd = (ctypes.c_float * 2)(float(scales[0]), float(scales[1]))
        rc = L.lib().devo_corr_forward_pyramid2(L.ptr(f1), L.ptr(d0[4]), L.ptr(d1[4]), L.ptr(c_), L.ptr(ii_), L.ptr(jj_),
                                                L.ptr(out), B, E, Np, pyramid[0].shape[1], C, P, hw, L.i64arr(d0[2] + d1[2]), cb,
                                                per * nl, nl, L.i64arr([0, 1]), int(radius), L.dtype_code(f1), L.ptr(order), cd,
                                                L.ptr(f1t), L.ptr(d0[5]), L.ptr(d1[5]), plan_kind(order, B * E), L.stream())
        if rc == 0:
            return out
        if rc != 3:                                             # DEVO_ERR_UNSUPPORTED: fall through to one launch per level
            L.check(rc, "cuda_corr.forward_pyramid")
    for lvl, (fm, s) in enumerate(zip(raw, scales)):
        forward_into(out, fmap1, fm, coords, ii, jj, radius, per * nl, nl, lvl, order=order, coord_div=s)
    return out


_cl_cache = {}             # (ptr, version, shape, strides, dtype) -> (source tensor [kept alive], channels-last copy); LRU
last_backward_path = None  # "product" | "segments" | "atomic": what the last backward() launched (devo_corr_backward_last_path)


def last_forward_path():
    """"dense-product" | "dense-product-groups" | "mfma4x4" | "staged" | "generic": the kernel the last forward lookup of this thread launched
    (devo_corr_forward_last_path); the slow ones also announce themselves once on stderr."""
    p = int(L.lib().devo_corr_forward_last_path())
    return ("dense-product", "mfma4x4", "staged", "generic", "dense-product-groups")[p] if 0 <= p < 5 else None


def _channels_last_copy(fmap2):
    """fmap2 [B, n, C, H, W] in any layout -> the same logical tensor with channels-last strides (a [B, n, H, W, C] buffer viewed
    as [B, n, C, H, W]), cached per version of the tensor with _fast_layout's key discipline: a training step calls backward() once
    per update iteration and level on the SAME pyramid tensors (enet.py:203-216), so the copy is made once per step and level."""
    key = (fmap2.data_ptr(), fmap2._version, tuple(fmap2.shape), tuple(fmap2.stride()), fmap2.dtype)
    hit = _cl_cache.pop(key, None)
    if hit is not None:
        _cl_cache[key] = hit
        return hit[1]
    for k in [k for k in _cl_cache if k[0] == key[0]]:
        del _cl_cache[k]
    while len(_cl_cache) >= BLOCKED_CACHE_ENTRIES:
        del _cl_cache[next(iter(_cl_cache))]
    cl = fmap2.detach().permute(0, 1, 3, 4, 2).contiguous().permute(0, 1, 4, 2, 3)
    _cl_cache[key] = (fmap2.detach(), cl)                 # (detached: pins the storage and shares the version counter, not the autograd graph)
    return cl


def _is_channels_last(fmap2):
    B, n, C, H, W = fmap2.shape
    st = fmap2.stride()
    return st[2] == 1 and st[4] == C and st[3] == W * C and st[1] >= H * W * C


def backward(fmap1, fmap2, coords, ii, jj, grad, radius):
    """correlation.cpp:59: the compiled binding's function when present, else the ctypes form below."""
    global last_backward_path
    N = _nat()
    if N is not None:
        out = N.cuda_corr.backward(fmap1, fmap2, coords, ii, jj, grad, int(radius))
        last_backward_path = N.cuda_corr.last_backward_path() or None
        return out
    return _backward_ctypes(fmap1, fmap2, coords, ii, jj, grad, radius)


def _backward_ctypes(fmap1, fmap2, coords, ii, jj, grad, radius):
    """correlation.cpp:59 -> [fmap1_grad, fmap2_grad] (fp32 only, like the reference's float grad accessor).
    fmap2_grad has fmap2's logical shape; for a plain NCHW fmap2 with C % 128 == 0 and >= NCHW_CONVERT_MIN_EDGES edges (what an
    unmodified enet.py hands over) it carries channels-last strides: the product form (corr_bwd_mfma.h) reads a cached channels-last
    copy of fmap2 and writes the gradient in that layout — autograd takes a gradient in any strides."""
    global last_backward_path
    fmap1, fmap2, coords, ii, jj = _prep(fmap1, fmap2, coords, ii, jj)
    L.require_gpu(grad)
    if fmap1.dtype != torch.float32:
        # the reference dispatches its backward over half / float / double with a FLOAT gradient accessor
        # (correlation_kernel.cu:146,280); here the kernel is fp32: other dtypes are computed in fp32 and cast back
        dt = fmap1.dtype
        d1, d2 = _backward_ctypes(fmap1.float(), fmap2.float(), coords, ii, jj, grad.float(), radius)
        return [d1.to(dt), d2.to(dt)]
    B, E = coords.shape[:2]
    P = coords.shape[3]
    _, Np, C = fmap1.shape[:3]
    n2, H2, W2 = fmap2.shape[1], fmap2.shape[3], fmap2.shape[4]
    grad = grad.float().contiguous()
    d1 = torch.empty_like(fmap1)
    cl = _is_channels_last(fmap2)
    if (not cl and C % 128 == 0 and B * E >= NCHW_CONVERT_MIN_EDGES and os.environ.get("DEVO_CORR_NCHW_DIRECT", "0") != "1"
            and int(L.lib().devo_corr_backward_workspace_bytes(B, E, Np, n2, C, int(radius), 1)) > 0):
        fmap2 = _channels_last_copy(fmap2)
        cl = True
    if cl:
        d2 = torch.empty(B, n2, H2, W2, C, dtype=fmap2.dtype, device=fmap2.device).permute(0, 1, 4, 2, 3)
        if fmap2.stride() != d2.stride():                     # (channels-last frames with a padded frame stride)
            d2 = torch.empty_strided(fmap2.shape, fmap2.stride(), dtype=fmap2.dtype, device=fmap2.device)
    else:
        d2 = torch.empty_strided(fmap2.shape, fmap2.stride(), dtype=fmap2.dtype, device=fmap2.device)
    span = 1 + sum((s - 1) * st for s, st in zip(fmap2.shape, fmap2.stride()))
    # scratch of the product form (torch's caching allocator); 0 for layouts that take the one-kernel atomic path
    nws = int(L.lib().devo_corr_backward_workspace_bytes(B, E, Np, n2, C, int(radius), 1 if cl else 0))
    ws = torch.empty(nws, dtype=torch.uint8, device=fmap1.device) if nws else None
    rc = L.lib().devo_corr_backward(L.ptr(fmap1), L.ptr(fmap2), L.ptr(coords), L.ptr(ii), L.ptr(jj), L.ptr(grad),
                                    L.ptr(d1), L.ptr(d2), B, E, Np, n2, C, P, H2, W2, L.i64arr(fmap2.stride()), span,
                                    int(radius), L.dtype_code(fmap1), L.ptr(ws) if ws is not None else None, nws, L.stream())
    L.check(rc, "cuda_corr.backward")
    last_backward_path = ("atomic", "segments", "product")[int(L.lib().devo_corr_backward_last_path())]
    return [d1, d2]


def patchify_forward(net, coords, radius):
    """correlation.cpp:61: net [B,C,H,W], coords [B,M,2] -> [patches [B,M,C,D,D]]"""
    N = _nat()
    if N is not None:
        return N.cuda_corr.patchify_forward(net, coords, int(radius))
    L.require_gpu(net, coords)
    B, M = coords.shape[:2]
    C, H, W = net.shape[1:]
    D = 2 * int(radius) + 2
    coords = coords.float().contiguous()
    out = torch.empty(B, M, C, D, D, dtype=net.dtype, device=net.device)
    rc = L.lib().devo_patchify_forward(L.ptr(net), L.ptr(coords), L.ptr(out), B, M, C, H, W, L.i64arr(net.stride()),
                                       int(radius), L.dtype_code(net), L.stream())
    L.check(rc, "cuda_corr.patchify_forward")
    return [out]


def patchify_backward(net, coords, gradient, radius):
    """correlation.cpp:62 -> [net_gradient [B,C,H,W]]"""
    N = _nat()
    if N is not None:
        return N.cuda_corr.patchify_backward(net, coords, gradient, int(radius))
    L.require_gpu(net, coords, gradient)
    B, M = coords.shape[:2]
    C, H, W = net.shape[1:]
    coords = coords.float().contiguous()
    if net.dtype == torch.float16:                    # scattered in fp32 (hardware float atomics), cast back like the forward's dtype
        return [patchify_backward(net.float(), coords, gradient.float(), radius)[0].half()]
    gradient = gradient.to(net.dtype).contiguous()
    # the gradient in net's own layout when that is a dense permutation (channels-last from the encoders' convolutions: no layout copy
    # on the way back into their backward), contiguous otherwise
    dense = net.numel() > 0 and 1 + sum((n - 1) * st for n, st in zip(net.shape, net.stride())) == net.numel()
    out = torch.empty_strided(net.shape, net.stride(), dtype=net.dtype, device=net.device) if dense else torch.empty(B, C, H, W, dtype=net.dtype, device=net.device)
    rc = L.lib().devo_patchify_backward(L.ptr(coords), L.ptr(gradient), L.ptr(out), B, M, C, H, W, L.i64arr(out.stride()), int(radius),
                                        L.dtype_code(net), L.stream())
    L.check(rc, "cuda_corr.patchify_backward")
    return [out]
