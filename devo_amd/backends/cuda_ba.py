"""HIP-backed module with the interface of the reference's `cuda_ba` extension (devo/fastba/ba.cpp:152-157):
forward (in-place bundle adjustment), neighbors, reproject — plus `transform`, the fused form of
devo/projective_ops.py:53-105 that DEVO.update really calls.  No CPU fallback."""
import ctypes
import torch
from .. import _lib as L


def _idx(*ts):
    return [t.long().contiguous() for t in ts]


def _nat():
    from . import native
    return native()


def workspace(E, Np, N, device):
    nbytes = L.lib().devo_ba_workspace_bytes(int(E), int(Np), int(N))
    if nbytes == 0:
        raise RuntimeError(f"cuda_ba: unsupported problem size (E={E}, Np={Np}, N={N}; at most 128 optimised poses)")
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def prepare(kk, n_patch_slots, n_opt, ws, plan=None):
    """Index half of cuda_ba.forward (ba_cuda.cu:435-437: unique patches, edges grouped by patch), which depends on
    `kk` only: call it early — e.g. on a side stream while the correlation lookup runs — and pass `prepared=True` to
    forward().  `n_patch_slots` = patches.shape[1], `n_opt` = t1 - t0; `ws` from workspace().
    plan=(buffer, n_frames, height[, width, l1]): the half-built locality plan of transform(..., plan_for=...) (width, l1 as given
    there: a group plan) is finished in the same
    launch (two independent single-workgroup kernels side by side); the buffer is then what cuda_corr.plan_finish returns."""
    L.require_gpu(kk, ws)
    kk = kk.long().contiguous()
    if plan is None:
        rc = L.lib().devo_ba_prepare(L.ptr(kk), kk.numel(), int(n_patch_slots), int(n_opt), L.ptr(ws), ws.numel(), L.stream())
    else:
        buf, n_frames, height = plan[:3]
        width, l1 = (tuple(plan[3:5]) + (0, 0))[:2]
        L.require_gpu(buf)
        groups = buf.numel() == 2 * kk.numel() + 2 + L.PLAN_TAIL and int(l1) >= 2
        if buf.dtype != torch.int32 or buf.numel() < 2 * kk.numel() + 2:
            raise RuntimeError("cuda_ba.prepare: plan must be the int32 buffer of transform(..., plan_for=...)")
        rc = L.lib().devo_ba_prepare_plan(L.ptr(kk), kk.numel(), int(n_patch_slots), int(n_opt), L.ptr(ws), ws.numel(),
                                          L.ptr(buf), int(n_frames), int(height), int(width) if groups else 0, int(l1) if groups else 0, L.stream())
    L.check(rc, "cuda_ba.prepare")
    prep_invalidate()                                        # the tables in `ws` are this kk's now, whatever forward()'s cache remembers
    return ws


# forward()'s prepared-table cache of the ctypes binding (the compiled binding has its own, csrc/bind.cpp): the index half of the BA depends
# on kk alone and DEVO runs the BA again and again on one graph — key = (kk's storage, version counter, E, patch slots, window size,
# workspace, stream); kk and the workspace are kept alive, so an equal key is the same storage with the same contents.
_prep = {"key": None, "keep": None, "hits": 0, "misses": 0}
_own_ws = {"key": None, "ws": None}
import os as _os
PREP_CACHE = _os.environ.get("DEVO_BA_PREP_CACHE", "1") != "0"


# The Update operator's patch-group tables as a DONOR of the BA's index tables (round 6): devo.py:311,337 hand the same kk to the operator and,
# right behind it, to fastba.BA — graph_tables() has then grouped exactly this edge list by patch, in a workspace of other sizes; forward() imports
# those tables with one launch (devo_ba_import_tables) instead of preparing them again with nine.  key = (kk's storage, version counter, E,
# stream); the donor keeps kk and its workspace alive, so an equal key is the same storage with the same contents.  _imported: what forward()
# last put into which BA workspace this way — any other preparation of a workspace (prepare(), a forward() on another kk, a capture) forgets it.
_donor = {"key": None, "ws": None, "sizes": None, "keep": None}
_imported = {"key": None, "keep": None, "n": 0}
SHARE_TABLES = _os.environ.get("DEVO_BA_SHARE_TABLES", "1") != "0"


def _kk_key(kk, stream_value):
    return (kk.data_ptr(), kk._version, kk.numel(), stream_value)


def offer_tables(kk, ws, n_patch_slots, n_opt):
    """`ws` holds devo_ba_prepare's tables of `kk` for (n_patch_slots, n_opt): forward() on the same kk may import them (see _donor)."""
    if SHARE_TABLES and kk.is_cuda and not kk.is_inference():
        st = L.stream()
        if L.lib().devo_stream_capturing(st) == 0:
            _donor.update(key=_kk_key(kk, st.value), ws=ws, sizes=(int(n_patch_slots), int(n_opt)), keep=kk)
            return
    _donor.update(key=None, ws=None, sizes=None, keep=None)


def import_stats():
    """Number of forward() calls that took their index tables from the Update operator's (devo_ba_import_tables)."""
    return _imported["n"]


def _try_import(kk0, E, Np, n_opt, ws, st):
    """-> True when `ws` holds kk0's tables for (Np, n_opt) through the donor (imported now or by the previous forward() on this workspace)."""
    if not SHARE_TABLES or _donor["key"] is None or kk0.is_inference() or _donor["key"] != _kk_key(kk0, st.value):
        return False
    if L.lib().devo_stream_capturing(st) != 0:
        return False
    want = (_donor["key"], id(_donor["ws"]), ws.data_ptr(), int(Np), int(n_opt))
    if _imported["key"] != want:
        src = _donor["ws"]
        rc = L.lib().devo_ba_import_tables(L.ptr(src), src.numel(), _donor["sizes"][0], _donor["sizes"][1], L.ptr(ws), ws.numel(), int(E), int(Np), int(n_opt), st)
        L.check(rc, "cuda_ba.forward (import of the Update operator's tables)")
        prep_invalidate()                                    # whatever the bindings' caches remember about this workspace is gone
        _imported["key"], _imported["keep"] = want, (ws, src)
        _imported["n"] += 1
    return True


def prep_invalidate():
    """Forget the prepared index tables forward() remembers (both bindings)."""
    _prep["key"] = _prep["keep"] = None
    _imported["key"] = _imported["keep"] = None
    N = _nat()
    if N is not None:
        N.cuda_ba._prep_invalidate()


def prep_stats():
    """(hits, misses) of forward()'s prepared-table cache in the active binding."""
    N = _nat()
    return tuple(N.cuda_ba._prep_stats()) if N is not None else (_prep["hits"], _prep["misses"])


def prepared_tables(ws, E, n_patch_slots, n_opt, sync=True):
    """(n_seg, kx, seg_start, perm) of a prepared workspace — the sorted unique patch ids of kk and its edges grouped by
    patch (the index work of ba_cuda.cu:435-437).  sync=True (tests): n_seg as a Python int, kx / seg_start cut to it — one host
    synchronisation; sync=False: n_seg as an int32 device tensor [1], kx / seg_start at their full length (entries beyond n_seg: seg_start = E)."""
    dev = ws.device
    m = min(int(E), int(n_patch_slots))
    n_seg = torch.empty(1, dtype=torch.int32, device=dev)
    kx = torch.empty(m, dtype=torch.int32, device=dev)
    seg = torch.empty(m + 1, dtype=torch.int32, device=dev)
    perm = torch.empty(int(E), dtype=torch.int32, device=dev)
    rc = L.lib().devo_ba_prepared_tables(L.ptr(ws), ws.numel(), int(E), int(n_patch_slots), int(n_opt), L.ptr(n_seg), L.ptr(kx),
                                         L.ptr(seg), L.ptr(perm), L.stream())
    L.check(rc, "cuda_ba.prepared_tables")
    if not sync:
        return n_seg, kx, seg, perm
    n = int(n_seg)
    return n, kx[:n], seg[:n + 1], perm


def forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, t0, t1, iterations, ws=None, status=None,
            prepared=False):
    """ba.cpp:153.  Mutates `poses` ([1,Nbuf,7]) and `patches` ([1,Np,3,P,P]) in place and returns []
    (devo/fastba/ba.py:7-8 passes poses.data; devo/devo.py:337 relies on the mutation).
    prepared=True: `ws` already holds the result of prepare() for this kk / t1 - t0.
    With the compiled binding present this is devo_amd._C.cuda_ba.forward (same arguments)."""
    imported = False
    if not prepared and ws is not None and _donor["key"] is not None and int(iterations) > 0 and kk.is_cuda and kk.numel() > 0:
        P_ = patches.shape[-1]
        imported = _try_import(kk, kk.numel(), patches.numel() // (3 * P_ * P_), int(t1) - int(t0), ws, L.stream())
    if not imported and not prepared:
        _imported["key"] = _imported["keep"] = None          # this call prepares its workspace itself: nothing imported survives in it
    prepared = bool(prepared) or imported
    N = _nat()
    if N is not None:
        return N.cuda_ba.forward(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, int(t0), int(t1), int(iterations), ws, status, bool(prepared))
    L.require_gpu(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk)
    for name, t in (("poses", poses), ("patches", patches)):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"cuda_ba.forward: {name} must be a contiguous float32 tensor (it is updated in place)")
    P = patches.shape[-1]
    Nbuf = poses.numel() // 7
    Np = patches.numel() // (3 * P * P)
    kk0 = kk                                                 # (the caller's tensor: what the prepared-table cache is keyed on)
    ii, jj, kk = _idx(ii, jj, kk)
    E = ii.numel()
    intrinsics = intrinsics.float().contiguous()
    target = target.float().contiguous()
    weight = weight.float().contiguous()
    lmbda = lmbda.float().reshape(-1).contiguous()
    if prepared and ws is None:
        raise RuntimeError("cuda_ba.forward: prepared=True needs the workspace that prepare() filled")
    st = L.stream()
    if ws is None:                                           # one live workspace per (size, device, stream), like the compiled binding
        wkey = (E, Np, int(t1) - int(t0), str(poses.device), st.value)
        if _own_ws["key"] != wkey:
            _own_ws["key"], _own_ws["ws"] = wkey, workspace(E, Np, int(t1) - int(t0), poses.device)
        ws = _own_ws["ws"]
    if prepared:
        _prep["key"] = _prep["keep"] = None
    elif PREP_CACHE and E > 0 and int(iterations) > 0:
        if L.lib().devo_stream_capturing(st) != 0:
            _prep["key"] = _prep["keep"] = None              # (a capture executes nothing; a replay rewrites the tables behind the cache's back)
        elif not kk0.is_inference():
            key = (kk0.data_ptr(), kk0._version, E, Np, int(t1) - int(t0), ws.data_ptr(), st.value)
            if _prep["key"] == key:
                prepared = True
                _prep["hits"] += 1
            else:
                _prep["key"], _prep["keep"] = key, (kk0, ws)
                _prep["misses"] += 1
    fn = L.lib().devo_ba_forward_prepared if prepared else L.lib().devo_ba_forward
    rc = fn(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(target), L.ptr(weight),
            L.ptr(lmbda), L.ptr(ii), L.ptr(jj), L.ptr(kk), E, Nbuf, Np, P, int(t0), int(t1),
            int(iterations), L.ptr(ws), ws.numel(), L.ptr(status), st)
    L.check(rc, "cuda_ba.forward")
    return []


def forward_delta(poses, patches, intrinsics, coords, delta, weight, lmbda, ii, jj, kk, t0, t1, iterations, ws, layout="2pp",
                  status=None, plan_next=None):
    """forward(..., prepared=True) with devo/devo.py:330 folded in: target = coords[..., P//2, P//2] + delta is formed inside
    the BA from the buffer transform() returned (`layout` as there) and the update operator's delta [1,E,2] — the same fp32
    addition, bit-identical results, one elementwise launch less.  `ws` must hold prepare()'s result.
    plan_next=(buffer, n_frames, height[, width, l1]): the half-built locality plan of transform(..., plan_for=...) — its ordering step
    rides on the first Gauss-Newton iteration's solver launch (devo_ba_forward_prepared_delta_plan); afterwards the buffer is what
    cuda_corr.plan_finish returns, for the NEXT update iteration's lookup (a plan only decides which edges run together)."""
    L.require_gpu(poses, patches, intrinsics, coords, delta, weight, lmbda, ii, jj, kk)
    for name, t in (("poses", poses), ("patches", patches)):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"cuda_ba.forward_delta: {name} must be a contiguous float32 tensor (it is updated in place)")
    if coords.dtype != torch.float32 or not coords.is_contiguous():
        raise RuntimeError("cuda_ba.forward_delta: coords must be the contiguous float32 tensor transform() returned")
    P = patches.shape[-1]
    Nbuf = poses.numel() // 7
    Np = patches.numel() // (3 * P * P)
    ii, jj, kk = _idx(ii, jj, kk)
    E = ii.numel()
    ctr = (P // 2) * (P + 1)
    if layout == "2pp" and coords.numel() == E * 2 * P * P:
        se, sc, off = 2 * P * P, P * P, ctr
    elif layout == "pp2" and coords.numel() == E * P * P * 2:
        se, sc, off = 2 * P * P, 1, 2 * ctr
    else:
        raise RuntimeError(f"cuda_ba.forward_delta: coords {tuple(coords.shape)} is not a '{layout}' buffer of {E} edges")
    intrinsics = intrinsics.float().contiguous()
    delta = delta.float().contiguous()
    weight = weight.float().contiguous()
    lmbda = lmbda.float().reshape(-1).contiguous()
    if delta.numel() != 2 * E:
        raise RuntimeError("cuda_ba.forward_delta: delta must hold 2 values per edge")
    if plan_next is not None:
        buf, n_frames, height = plan_next[:3]
        width, l1 = (tuple(plan_next[3:5]) + (0, 0))[:2]
        L.require_gpu(buf)
        if buf.dtype != torch.int32 or buf.numel() < 2 * E + 2:
            raise RuntimeError("cuda_ba.forward_delta: plan_next must be the int32 buffer of transform(..., plan_for=...)")
        groups = buf.numel() == 2 * E + 2 + L.PLAN_TAIL and int(l1) >= 2
        rc = L.lib().devo_ba_forward_prepared_delta_plan(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(coords), se, sc, off,
                                                         L.ptr(delta), L.ptr(weight), L.ptr(lmbda), L.ptr(ii), L.ptr(jj), L.ptr(kk), E,
                                                         Nbuf, Np, P, int(t0), int(t1), int(iterations), L.ptr(ws), ws.numel(),
                                                         L.ptr(status), L.ptr(buf), int(n_frames), int(height), int(width) if groups else 0,
                                                         int(l1) if groups else 0, L.stream())
        L.check(rc, "cuda_ba.forward_delta")
        return []
    rc = L.lib().devo_ba_forward_prepared_delta(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(coords), se, sc, off,
                                                L.ptr(delta), L.ptr(weight), L.ptr(lmbda), L.ptr(ii), L.ptr(jj), L.ptr(kk), E,
                                                Nbuf, Np, P, int(t0), int(t1), int(iterations), L.ptr(ws), ws.numel(),
                                                L.ptr(status), L.stream())
    L.check(rc, "cuda_ba.forward_delta")
    return []


def neighbors(ii, jj):
    """ba.cpp:154 -> [ix, jx] (int64, on the GPU); no device<->host round trip (the reference does five)."""
    N = _nat()
    if N is not None:
        return N.cuda_ba.neighbors(ii, jj)
    L.require_gpu(ii, jj)
    ii, jj = _idx(ii, jj)
    E = ii.numel()
    ix = torch.empty(E, dtype=torch.int64, device=ii.device)
    jx = torch.empty(E, dtype=torch.int64, device=ii.device)
    ws = torch.empty(L.lib().devo_neighbors_workspace_bytes(E), dtype=torch.uint8, device=ii.device)
    rc = L.lib().devo_ba_neighbors(L.ptr(ii), L.ptr(jj), L.ptr(ix), L.ptr(jx), E, L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, "cuda_ba.neighbors")
    return [ix, jx]


_offsets = {}


def table_offsets(E, n_patch_slots, n_opt):
    """(n_seg, kx, seg_start, perm) byte offsets into a prepared workspace of these sizes + the length of kx (devo_ba_table_offsets)."""
    key = (int(E), int(n_patch_slots), int(n_opt))
    o = _offsets.get(key)
    if o is None:
        buf = (ctypes.c_size_t * 5)()
        L.check(L.lib().devo_ba_table_offsets(*key, ctypes.cast(buf, ctypes.c_void_p)), "cuda_ba.table_offsets")
        if len(_offsets) > 64:
            _offsets.clear()
        o = _offsets[key] = tuple(int(v) for v in buf)
    return o


def table_views(ws, E, n_patch_slots, n_opt):
    """(n_seg i32 [1], kx, seg_start, perm) as int32 VIEWS of a prepared workspace (no copies; valid until the workspace is prepared again)."""
    o_n, o_kx, o_seg, o_perm, m = table_offsets(E, n_patch_slots, n_opt)
    v = lambda off, n: ws[off:off + 4 * n].view(torch.int32)
    return v(o_n, 1), v(o_kx, m), v(o_seg, m + 1), v(o_perm, int(E))


def graph_tables(ii, jj, kk, bound=1 << 20):
    """The Update operator's tables of one edge list (devo/enet.py:86-95) from ONE library call (devo_upd_graph_tables): ix, jx =
    neighbors(kk, jj); the edges grouped by patch kk; the edges grouped by frame pair (ii, jj).  Returns
    (ix, jx, (n_seg, seg_start, perm) by patch, (n_seg, seg_start, perm) by pair): int32 device views of two workspaces this call
    allocates (n_seg: [1], the group count stays on the device).  kk in [0, bound), (frames in the window)^2 <= bound."""
    L.require_gpu(ii, jj, kk)
    kk_given = kk                                            # (the caller's tensor: what the donor is keyed on)
    ii, jj, kk = _idx(ii, jj, kk)
    E, dev = kk.numel(), kk.device
    if E == 0:
        raise RuntimeError("cuda_ba.graph_tables: empty edge list")
    nbytes = int(L.lib().devo_ba_workspace_bytes(E, int(bound), 0))
    ws = torch.empty(2, nbytes, dtype=torch.uint8, device=dev)
    nb = torch.empty(2, E, dtype=torch.int64, device=dev)
    key = torch.empty(E + 2, dtype=torch.int64, device=dev)
    rc = L.lib().devo_upd_graph_tables(L.ptr(ii), L.ptr(jj), L.ptr(kk), E, int(bound), L.ptr(ws[0]), nbytes, L.ptr(ws[1]), nbytes,
                                       L.ptr(key), L.ptr(nb[0]), L.ptr(nb[1]), L.stream())
    L.check(rc, "cuda_ba.graph_tables")
    tk, tp = table_views(ws[0], E, bound, 0), table_views(ws[1], E, bound, 0)
    offer_tables(kk_given, ws[0], bound, 0)                  # the patch groups: what fastba.BA needs for this kk right behind the operator
    return nb[0], nb[1], (tk[0], tk[2], tk[3]), (tp[0], tp[2], tp[3])


def reproject(poses, patches, intrinsics, ii, jj, kk):
    """ba.cpp:155 -> coords [1, E, 2, P, P] (no depth clamp, ba_cuda.cu:368-418)."""
    N = _nat()
    if N is not None:
        return N.cuda_ba.reproject(poses, patches, intrinsics, ii, jj, kk)
    L.require_gpu(poses, patches, intrinsics, ii, jj, kk)
    P = patches.shape[-1]
    ii, jj, kk = _idx(ii, jj, kk)
    E = ii.numel()
    poses = poses.float().contiguous()
    patches = patches.float().contiguous()
    intrinsics = intrinsics.float().contiguous()
    coords = torch.empty(1, E, 2, P, P, dtype=torch.float32, device=poses.device)
    rc = L.lib().devo_ba_reproject(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(ii), L.ptr(jj), L.ptr(kk),
                                   L.ptr(coords), E, P, L.stream())
    L.check(rc, "cuda_ba.reproject")
    return coords


def transform(poses, patches, intrinsics, ii, jj, kk, depth=False, valid=False, jacobian=False, tonly=False,
              layout="pp2", plan_for=None):
    """Fused projective transform with the semantics of devo/projective_ops.py:53-105 (batch 1, no autograd).
    layout "pp2": coords [1,E,P,P,2|3] as the reference returns; "2pp": [1,E,2,P,P] (devo/devo.py:223).
    plan_for=(n_frames, height, radius[, width, l1]): also start the lookup's locality plan for these coordinates (the kernel
    emits the plan bins while it holds them; width, l1 = 4: GROUP plan, see cuda_corr.plan); the half-built plan buffer is
    returned LAST — finish it with cuda_corr.plan_finish(buffer, jj, n_frames, height, radius[, width=, l1=]) or
    prepare(..., plan=(buffer, n_frames, height[, width, l1]))."""
    if not (depth or valid or jacobian or tonly) and plan_for is None:
        N = _nat()
        if N is not None:                                      # coordinates only (DEVO.reproject, devo.py:218-223): the compiled binding's short form
            return N.cuda_ba.transform_coords(poses, patches, intrinsics, ii, jj, kk, layout == "2pp")
    L.require_gpu(poses, patches, intrinsics, ii, jj, kk)
    P = patches.shape[-1]
    ii, jj, kk = _idx(ii, jj, kk)
    E = ii.numel()
    dev = poses.device
    poses = poses.float().contiguous()
    patches = patches.float().contiguous()
    intrinsics = intrinsics.float().contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    c_pp2 = torch.empty(1, E, P, P, 3 if depth else 2, **f32) if layout == "pp2" else None
    c_2pp = torch.empty(1, E, 2, P, P, **f32) if layout == "2pp" else None
    v = torch.empty(1, E, **f32) if (valid or jacobian) else None
    Ji = torch.empty(1, E, 2, 6, **f32) if jacobian else None
    Jj = torch.empty(1, E, 2, 6, **f32) if jacobian else None
    Jz = torch.empty(1, E, 2, 1, **f32) if jacobian else None
    flags = (1 if depth else 0) | (2 if tonly else 0)
    plan, pf = None, (0, 0, 0, 0, 0)
    if plan_for is not None:
        from . import cuda_corr
        pf = (tuple(int(x) for x in plan_for) + (0, 0))[:5]
        groups = pf[4] >= 2 and cuda_corr.group_plan_supported(1, pf[0], pf[1], pf[3], pf[4], pf[2])
        if not groups:
            pf = pf[:3] + (0, 0)                              # (geometries without a group plan: an edge plan)
        plan = cuda_corr.plan_buffer(E, dev, groups)
    rc = L.lib().devo_transform(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(ii), L.ptr(jj), L.ptr(kk),
                                L.ptr(c_pp2), L.ptr(c_2pp), L.ptr(v), L.ptr(Ji), L.ptr(Jj), L.ptr(Jz), E, P, flags,
                                L.ptr(plan), pf[0], pf[1], pf[2], pf[3], pf[4], L.stream())
    L.check(rc, "cuda_ba.transform")
    c = c_pp2 if layout == "pp2" else c_2pp
    out = (c, v, (Ji, Jj, Jz)) if jacobian else ((c, v) if valid else c)
    if plan_for is None:
        return out
    return (out + (plan,)) if isinstance(out, tuple) else (out, plan)


def solve_terms(terms, lmbda, ii, jj, kk, n_patch_slots, t0, n_opt, ep, status=None):
    """devo_ba_solve_terms: one Gauss-Newton step of devo/ba.py:108-170 from given edge terms [E,30] -> (dX [6 n_opt], dZ [Np], ws);
    `ws` holds what solve_terms_backward needs."""
    L.require_gpu(terms, lmbda, ii, jj, kk)
    ii, jj, kk = _idx(ii, jj, kk)
    E = ii.numel()
    terms = terms.float().contiguous()
    lmbda = lmbda.float().reshape(-1).contiguous()
    ws = workspace(E, int(n_patch_slots), int(n_opt), terms.device)
    out = torch.empty(6 * int(n_opt) + int(n_patch_slots), dtype=torch.float32, device=terms.device)     # dX | dZ: one fill clears both
    dX, dZ = out[:6 * int(n_opt)], out[6 * int(n_opt):]
    rc = L.lib().devo_ba_solve_terms(L.ptr(terms), L.ptr(lmbda), L.ptr(ii), L.ptr(jj), L.ptr(kk), E, int(n_patch_slots), int(t0), int(n_opt),
                                     float(ep), L.ptr(ws), ws.numel(), L.ptr(dX), L.ptr(dZ), L.ptr(status), L.stream())
    L.check(rc, "cuda_ba.solve_terms")
    return dX, dZ, ws


def solve_terms_backward(terms, ii, jj, kk, n_patch_slots, t0, n_opt, ws, g_dX, g_dZ):
    """devo_ba_solve_terms_backward -> g_terms [E,30]"""
    ii, jj, kk = _idx(ii, jj, kk)
    E = ii.numel()
    terms = terms.float().contiguous()
    g_dX = g_dX.float().contiguous()
    g_dZ = g_dZ.float().contiguous()
    g = torch.empty(E, 30, dtype=torch.float32, device=terms.device)
    rc = L.lib().devo_ba_solve_terms_backward(L.ptr(terms), L.ptr(ii), L.ptr(jj), L.ptr(kk), E, int(n_patch_slots), int(t0), int(n_opt), L.ptr(ws),
                                              ws.numel(), L.ptr(g_dX), L.ptr(g_dZ), L.ptr(g), L.stream())
    L.check(rc, "cuda_ba.solve_terms_backward")
    return g


def edge_terms(coords, valid, Ji, Jj, Jz, target, weight, bounds):
    """devo_ba_edge_terms (devo/ba.py:95-106): transform's outputs + targets / weights / bounds -> (terms [E,30], gate [E])"""
    L.require_gpu(coords, valid, Ji, Jj, Jz, target, weight)
    E, P = coords.shape[1], coords.shape[2]
    c = lambda t: t.float().contiguous()
    coords, valid, Ji, Jj, Jz, target, weight = (c(t) for t in (coords, valid, Ji, Jj, Jz, target, weight))
    terms = torch.empty(E, 30, dtype=torch.float32, device=coords.device)
    gate = torch.empty(E, dtype=torch.float32, device=coords.device)
    b = (ctypes.c_float * 4)(*[float(v) for v in bounds])
    rc = L.lib().devo_ba_edge_terms(L.ptr(coords), L.ptr(valid), L.ptr(Ji), L.ptr(Jj), L.ptr(Jz), L.ptr(target), L.ptr(weight), b, E, P,
                                    L.ptr(terms), L.ptr(gate), L.stream())
    L.check(rc, "cuda_ba.edge_terms")
    return terms, gate


def edge_terms_backward(g_terms, gate, P):
    """devo_ba_edge_terms_backward -> (g_coords [1,E,P,P,2], g_target [1,E,2], g_weight [1,E,2], g_Ji [1,E,2,6], g_Jj [1,E,2,6], g_Jz [1,E,2,1])"""
    E, dev = gate.numel(), gate.device
    g_terms = g_terms.float().contiguous()
    new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
    gc, gt, gw, gi, gj, gz = new(1, E, P, P, 2), new(1, E, 2), new(1, E, 2), new(1, E, 2, 6), new(1, E, 2, 6), new(1, E, 2, 1)
    rc = L.lib().devo_ba_edge_terms_backward(L.ptr(g_terms), L.ptr(gate), E, int(P), L.ptr(gc), L.ptr(gt), L.ptr(gw), L.ptr(gi), L.ptr(gj), L.ptr(gz),
                                             L.stream())
    L.check(rc, "cuda_ba.edge_terms_backward")
    return gc, gt, gw, gi, gj, gz


def transform_vjp(poses, patches, intrinsics, ii, jj, kk, g_coords, g_J, depth=False, tonly=False):
    """devo_transform_vjp: adjoint of transform(layout="pp2") -> (g_poses [1,Nbuf,7], g_patches like patches).
    g_coords [1,E,P,P,2|3] or None; g_J = (g_Ji, g_Jj, g_Jz) or None (entries may be None, g_Jj must be there when any is)."""
    L.require_gpu(poses, patches, intrinsics, ii, jj, kk)
    P = patches.shape[-1]
    ii, jj, kk = _idx(ii, jj, kk)
    E = ii.numel()
    poses = poses.float().contiguous()
    patches = patches.float().contiguous()
    intrinsics = intrinsics.float().contiguous()
    c = lambda t: None if t is None else t.float().contiguous()
    g_coords = c(g_coords)
    gJi, gJj, gJz = (c(t) for t in g_J) if g_J is not None else (None, None, None)
    if (gJi is not None or gJz is not None) and gJj is None:
        gJj = torch.zeros(E, 2, 6, dtype=torch.float32, device=poses.device)
    gp = torch.empty_like(poses)
    gq = torch.empty_like(patches)
    flags = (1 if depth else 0) | (2 if tonly else 0)
    rc = L.lib().devo_transform_vjp(L.ptr(poses), L.ptr(patches), L.ptr(intrinsics), L.ptr(ii), L.ptr(jj), L.ptr(kk), L.ptr(g_coords),
                                    L.ptr(gJi), L.ptr(gJj), L.ptr(gJz), E, poses.numel() // 7, patches.numel() // (3 * P * P), P, flags,
                                    L.ptr(gp), L.ptr(gq), L.stream())
    L.check(rc, "cuda_ba.transform_vjp")
    return gp, gq


def last_path():
    """"accumulate:<register | lds | global> solve:<chain | lds | global>": the kernels the last forward() of this thread ran (devo_ba_last_path)."""
    p = int(L.lib().devo_ba_last_path())
    return None if p < 0 else f"accumulate:{('register', 'lds', 'global')[p & 3]} solve:{('chain', 'lds', 'global')[(p >> 2) & 3]}"
