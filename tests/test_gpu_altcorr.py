"""GPU parity of altcorr (cuda_corr.*) against the CPU oracle: the correlation lookup (channels-last fast
path, generic-stride path, fp32/fp16/fp64, r=3 and r=5, out-of-bounds / negative / integer / widely spread
coordinates), its backward, the fused pyramid output, and patchify forward/backward.
Tolerance: 1e-4 relative to the output scale for fp32 (north_star); 2e-3 for fp16 storage."""
import os
import subprocess
import sys
import pytest
import torch
from oracle import altcorr as A
from util import assert_rel, channels_last5

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _case(n=4, Np=20, C=128, H=30, W=40, E=200, R=3, seed=0, spread=1.0, jitter=True):
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(1, Np, C, 3, 3, generator=g) / 4
    f2 = torch.randn(1, n, C, H, W, generator=g) / 4
    base = torch.stack([torch.rand(E, generator=g) * (W + 12) - 6, torch.rand(E, generator=g) * (H + 12) - 6], 1)
    oy, ox = torch.meshgrid(torch.arange(3.) - 1, torch.arange(3.) - 1, indexing="ij")
    off = torch.stack([ox, oy], 0)                                    # [2,3,3]: x offsets vary along j0
    scale = spread * (0.6 + 0.9 * torch.rand(E, 1, 1, 1, generator=g))
    coords = base[:, :, None, None] + scale * off[None]
    if jitter:
        coords = coords + 0.2 * torch.randn(coords.shape, generator=g)
    coords[0] = coords[0].round()                                     # an exactly-integer edge (dx = dy = 0)
    coords[1] = -50.0 + off                                           # a fully out-of-bounds edge
    ii = torch.randint(0, Np, (E,), generator=g)
    jj = torch.randint(0, n, (E,), generator=g)
    return f1, f2, coords[None].contiguous(), ii, jj, R


def _run(f1, f2, coords, ii, jj, R, layout="cl", dtype=torch.float32):
    from devo_amd.backends import cuda_corr
    f1d = f1.to(DEV, dtype)
    f2d = f2.to(DEV, dtype)
    if layout == "cl":
        f2d = channels_last5(f2d)
    elif layout == "blk8":
        from devo_amd import altcorr
        f2d = altcorr.channel_blocked(f2d, 8)
    out, = cuda_corr.forward(f1d, f2d, coords.to(DEV), ii.to(DEV), jj.to(DEV), R)
    return out


@pytest.mark.parametrize("layout", ["cl", "nchw", "blk8"])
@pytest.mark.parametrize("R", [3, 5])
def test_forward_fp32(layout, R):
    c = _case(R=R, seed=R)
    ref = A.corr_forward(*c)
    got = _run(*c, layout=layout)
    assert got.shape == ref.shape and got.is_contiguous()
    assert_rel(got, ref, 1e-4, f"corr fwd {layout} R={R}")
    assert torch.count_nonzero(got[0, 1]) == 0                         # out-of-bounds taps are exactly 0


def test_forward_wide_spread_uses_fallback_path():
    """patch pixels far apart (bounding box > 512 positions): the per-tap path of the fast kernel."""
    c = _case(seed=7, spread=9.0, E=64)
    assert_rel(_run(*c, layout="cl"), A.corr_forward(*c), 1e-4, "wide spread")
    c = _case(seed=8, spread=3.5, E=64)                                 # several 128-position chunks
    assert_rel(_run(*c, layout="cl"), A.corr_forward(*c), 1e-4, "multi-chunk")


def test_dtype_coverage_of_the_reference_dispatch():
    """AT_DISPATCH_FLOATING_TYPES_AND_HALF (correlation_kernel.cu:203,247,299,320): fp64 lookups are fp64-accurate (fp64
    accumulation in the strided kernel), the backward passes accept fp16 tensors (computed in fp32, cast back)."""
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(E=40, C=32, seed=31)
    d = lambda t: t.to(DEV)
    ref = A.corr_forward(f1.double(), f2.double(), coords, ii, jj, R)
    got, = cuda_corr.forward(d(f1).double(), d(f2).double(), d(coords), d(ii), d(jj), R)
    assert got.dtype == torch.float64
    assert_rel(got, ref, 1e-12, "fp64 lookup")
    g = torch.randn(1, 40, 7, 7, 3, 3, generator=torch.Generator().manual_seed(1))
    a1, a2 = cuda_corr.backward(d(f1), d(f2), d(coords), d(ii), d(jj), d(g), R)
    h1, h2 = cuda_corr.backward(d(f1).half(), d(f2).half(), d(coords), d(ii), d(jj), d(g).half(), R)
    assert h1.dtype == h2.dtype == torch.float16
    assert_rel(h1.float(), a1, 5e-3, "fp16 d fmap1"); assert_rel(h2.float(), a2, 5e-3, "fp16 d fmap2")
    net = torch.randn(1, 8, 12, 16, generator=torch.Generator().manual_seed(2))
    pc = torch.tensor([[[3.0, 4.0], [10.0, 7.0], [0.0, 0.0]]])
    pg = torch.randn(1, 3, 8, 4, 4, generator=torch.Generator().manual_seed(3))
    b32, = cuda_corr.patchify_backward(d(net), d(pc), d(pg), 1)
    b16, = cuda_corr.patchify_backward(d(net).half(), d(pc), d(pg).half(), 1)
    assert b16.dtype == torch.float16
    assert_rel(b16.float(), b32, 5e-3, "fp16 patchify backward")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_nchw_pyramid_goes_through_a_cached_blocked_copy(dtype):
    """What an unmodified devo.py hands over: an NCHW ring buffer that is rewritten in place, one frame per step
    (devo/devo.py:71-83, 210-217).  From 1024 edges on the lookup converts it once per VERSION of the tensor into a
    channel-blocked copy: same bits as the explicit blocked layout, the copy is reused while the tensor is untouched, and an
    in-place frame update is seen."""
    from devo_amd import altcorr
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(E=1500, seed=21)
    d = lambda t: t.to(DEV)
    ring = d(f2).to(dtype)                                              # NCHW, like self.fmap1_
    g = d(f1).to(dtype)
    pyr = [ring, torch.nn.functional.avg_pool2d(ring[0].float(), 2, 2)[None].to(dtype)]
    call = lambda: cuda_corr.forward_pyramid(g, pyr, d(coords), d(ii), d(jj), R, (1, 2))
    ref = lambda: cuda_corr.forward_pyramid(g, [altcorr.channel_blocked(p_.clone(), 8) for p_ in pyr], d(coords), d(ii), d(jj), R, (1, 2))
    a = call()
    assert torch.equal(a, ref())
    n_cached = cuda_corr.cached_levels()
    assert n_cached >= 2 and torch.equal(call(), a) and cuda_corr.cached_levels() == n_cached      # cache hit: no new copy
    ring[:, 1] = ring[:, 1] * -0.5                                     # the per-frame in-place update of the ring (version bump)
    b = call()
    assert not torch.equal(a, b) and torch.equal(b, ref())
    small = cuda_corr.forward(g, ring, d(coords)[:, :64], d(ii)[:64], d(jj)[:64], R)[0]      # short edge lists read NCHW directly
    assert_rel(small, A.corr_forward(f1.to(dtype).float(), ring.float().cpu(), coords[:, :64], ii[:64], jj[:64], R), 2e-3 if dtype == torch.float16 else 1e-4, "nchw direct")


def test_channel_blocked_layout_is_bit_identical_and_checked():
    """[B,n,C/8,H,W,8] storage feeds the same kernel through other addresses: results are bit-identical to
    channels-last (split boxes and the opt-in LDS-direct kernel's layouts included); bad block sizes are rejected."""
    from devo_amd import altcorr
    from devo_amd.backends import cuda_corr
    for seed, spread in ((11, 1.0), (12, 3.5), (13, 9.0)):
        c = _case(seed=seed, spread=spread, E=96)
        assert torch.equal(_run(*c, layout="blk8"), _run(*c, layout="cl"))
    f1, f2, coords, ii, jj, R = _case(E=8)
    h = lambda t: t.to(DEV).half()
    with pytest.raises(RuntimeError):                                 # the staged kernel (fp16) reads 8-channel blocks only
        cuda_corr.forward(h(f1), altcorr.channel_blocked(h(f2), 4), coords.to(DEV), ii.to(DEV), jj.to(DEV), R)
    if os.environ.get("DEVO_CORR_MFMA", "1") != "0":                  # the matrix-core kernel: blocks of 4, 8 or 16 channels
        c = _case(seed=15, spread=3.5, E=96)
        for cb in (4, 16):
            out, = cuda_corr.forward(c[0].to(DEV), altcorr.channel_blocked(c[1].to(DEV), cb), c[2].to(DEV), c[3].to(DEV), c[4].to(DEV), c[5])
            assert torch.equal(out, _run(*c, layout="cl")), cb
        for cb in (16, 32):                                           # fp16 storage: blocks of 8, 16 or 32 channels
            out, = cuda_corr.forward(h(c[0]), altcorr.channel_blocked(h(c[1]), cb), c[2].to(DEV), c[3].to(DEV), c[4].to(DEV), c[5])
            assert torch.equal(out, _run(*c, layout="cl", dtype=torch.float16)), cb
    else:
        with pytest.raises(RuntimeError):
            cuda_corr.forward(f1.to(DEV), altcorr.channel_blocked(f2.to(DEV), 4), coords.to(DEV), ii.to(DEV), jj.to(DEV), R)
    with pytest.raises(RuntimeError):
        cuda_corr.forward(f1.to(DEV).double(), altcorr.channel_blocked(f2.to(DEV).double(), 8), coords.to(DEV), ii.to(DEV), jj.to(DEV), R)
    c = _case(seed=14, E=96)                                          # fp16 storage, blocked == channels-last
    assert torch.equal(_run(*c, layout="blk8", dtype=torch.float16), _run(*c, layout="cl", dtype=torch.float16))


def test_forward_fp16_and_fp64():
    c = _case(seed=11)
    ref = A.corr_forward(c[0].half().float(), c[1].half().float(), *c[2:])
    assert_rel(_run(*c, layout="cl", dtype=torch.float16), ref, 2e-3, "fp16 cl")
    assert_rel(_run(*c, layout="nchw", dtype=torch.float16), ref, 2e-3, "fp16 nchw")
    assert_rel(_run(*c, layout="nchw", dtype=torch.float64), A.corr_forward(*c), 1e-6, "fp64")


def test_small_channel_count_goes_generic():
    c = _case(C=24, seed=13)
    assert_rel(_run(*c, layout="cl"), A.corr_forward(*c), 1e-4, "C=24")


def test_fused_pyramid_equals_stack():
    """devo.py:215-217: torch.stack([corr1, corr2], -1).view(1, E, -1)"""
    from devo_amd import altcorr
    f1, f2, coords, ii, jj, R = _case(H=32, W=48, seed=17)
    f2b = torch.nn.functional.avg_pool2d(f2[0], 4, 4)[None]
    pyr = [channels_last5(f2.to(DEV)), channels_last5(f2b.to(DEV))]
    args = (coords.to(DEV), ii.to(DEV), jj.to(DEV))
    fused = altcorr.corr_pyramid(f1.to(DEV), pyr, *args, radius=R, scales=(1, 4))
    c1 = altcorr.corr(f1.to(DEV), pyr[0], args[0] / 1, args[1], args[2], R)
    c2 = altcorr.corr(f1.to(DEV), pyr[1], args[0] / 4, args[1], args[2], R)
    assert torch.equal(fused, torch.stack([c1, c2], -1).view(1, len(ii), -1))
    ref = torch.stack([A.corr_forward(f1, f2, coords, ii, jj, R), A.corr_forward(f1, f2b, coords / 4, ii, jj, R)], -1)
    assert_rel(fused, ref.view(1, len(ii), -1), 1e-4, "pyramid")


def _product_path():
    """what a channels-last C % 128 == 0 backward takes: the product form, unless test_other_kernels_stay_covered's sub-process asked for another"""
    return "atomic" if os.environ.get("DEVO_CORR_BWD_ATOMIC") else "segments" if os.environ.get("DEVO_CORR_BWD_SEG") else "product"


@pytest.mark.parametrize("layout", ["cl", "nchw"])
def test_backward_fp32(layout):
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(n=3, Np=12, C=64, H=20, W=24, E=60, seed=19)
    g = torch.randn(1, 60, 7, 7, 3, 3, generator=torch.Generator().manual_seed(1))
    r1, r2 = A.corr_backward(f1, f2, coords, ii, jj, g, R)
    f2d = f2.to(DEV)
    if layout == "cl":
        f2d = channels_last5(f2d)
    d1, d2 = cuda_corr.backward(f1.to(DEV), f2d, coords.to(DEV), ii.to(DEV), jj.to(DEV), g.to(DEV), R)
    # C = 64: never the product form; channels-last under DEVO_CORR_BWD_SEG=1 (test_other_kernels_stay_covered): the tile kernel
    want = "segments" if (os.environ.get("DEVO_CORR_BWD_SEG") and layout == "cl") else "atomic"
    assert d2.stride() == f2d.stride() and cuda_corr.last_backward_path == want
    assert_rel(d1, r1, 1e-4, "d_fmap1")
    assert_rel(d2, r2, 1e-4, "d_fmap2")


@pytest.mark.parametrize("C,R,B,E,H,W", [(128, 3, 1, 300, 20, 24), (256, 3, 1, 120, 12, 20), (128, 1, 1, 90, 20, 24), (128, 5, 1, 90, 33, 41),
                                          (128, 3, 2, 80, 20, 24), (128, 0, 1, 50, 9, 9)])
def test_backward_product_form(C, R, B, E, H, W):
    """channels-last fmap2 with C % 128 == 0 (DEVO: C = 128): d_fmap1 per edge and d_fmap2 per 8 x 8 frame tile as products on the fp32
    matrix cores (corr_bwd_mfma.h) — no atomics on d_fmap2, every position of every frame written exactly once (frames and tiles no edge
    touches included: the result tensor is NOT zeroed first).  Frame sizes that are not multiples of the tile, radii 0..5, batch 2,
    windows partly and wholly outside the frame, many edges per frame (several scan rounds of 256 windows)."""
    from devo_amd.backends import cuda_corr
    g = torch.Generator().manual_seed(1000 + C + R + B)
    n, Np, D = 4, 10, 2 * R + 1
    f1 = torch.randn(B, Np, C, 3, 3, generator=g) / 4
    f2 = torch.randn(B, n, C, H, W, generator=g) / 4
    base = torch.stack([torch.rand(B, E, generator=g) * (W + 12) - 6, torch.rand(B, E, generator=g) * (H + 12) - 6], 2)
    oy, ox = torch.meshgrid(torch.arange(3.) - 1, torch.arange(3.) - 1, indexing="ij")
    coords = (base[..., None, None] + 1.3 * torch.stack([ox, oy], 0) + 0.2 * torch.randn(B, E, 2, 3, 3, generator=g)).contiguous()
    coords[:, 0] = coords[:, 0].round()
    coords[:, 1] = -40.0
    ii = torch.randint(0, Np, (E,), generator=g)
    jj = torch.randint(0, n - 1, (E,), generator=g)                 # the last frame gets no edge at all
    grad = torch.randn(B, E, D, D, 3, 3, generator=g)
    r1, r2 = A.corr_backward(f1, f2, coords, ii, jj, grad, R)
    f2d = channels_last5(f2.to(DEV))
    d1, d2 = cuda_corr.backward(f1.to(DEV), f2d, coords.to(DEV), ii.to(DEV), jj.to(DEV), grad.to(DEV), R)
    assert cuda_corr.last_backward_path == _product_path()
    assert d2.stride() == f2d.stride() and torch.count_nonzero(d2[:, n - 1]) == 0
    assert_rel(d1, r1, 1e-4, "d_fmap1 (product form)")
    assert_rel(d2, r2, 1e-4, "d_fmap2 (product form)")


def test_backward_product_form_four_wave_tiles_with_empty_passes():
    """corr_bwd_frame_kernel<4> (8 x 8 tiles, what DEVO's level 0 takes in training): >= 2048 tiles in the launch, 600 edges = 5400
    windows per frame = three scan passes of 2304, all of them in the left part of the frame — every tile to the right keeps nothing in
    ANY pass (consecutive passes without a list barrier: the counts are double-buffered by pass parity), tiles at the cluster's border
    keep something in some passes only."""
    from devo_amd.backends import cuda_corr
    g = torch.Generator().manual_seed(77)
    B, n, Np, C, H, W, R, per = 1, 4, 40, 128, 128, 256, 3, 600
    E = per * (n - 1)
    f1 = torch.randn(B, Np, C, 3, 3, generator=g) / 4
    f2 = torch.randn(B, n, C, H, W, generator=g) / 4
    base = torch.stack([torch.rand(B, E, generator=g) * 110 - 6, torch.rand(B, E, generator=g) * (H + 12) - 6], 2)
    oy, ox = torch.meshgrid(torch.arange(3.) - 1, torch.arange(3.) - 1, indexing="ij")
    coords = (base[..., None, None] + 1.3 * torch.stack([ox, oy], 0) + 0.2 * torch.randn(B, E, 2, 3, 3, generator=g)).contiguous()
    ii = torch.randint(0, Np, (E,), generator=g)
    jj = torch.arange(E) % (n - 1)                                   # the last frame gets no edge at all
    grad = torch.randn(B, E, 7, 7, 3, 3, generator=g)
    r1, r2 = A.corr_backward(f1, f2, coords, ii, jj, grad, R)
    f2d = channels_last5(f2.to(DEV))
    for _ in range(3):                                               # (a race shows up as a hang or as a wrong tile in SOME run)
        d1, d2 = cuda_corr.backward(f1.to(DEV), f2d, coords.to(DEV), ii.to(DEV), jj.to(DEV), grad.to(DEV), R)
        assert cuda_corr.last_backward_path == _product_path()
        assert torch.count_nonzero(d2[:, n - 1]) == 0 and torch.count_nonzero(d2[..., 130:]) == 0
        assert_rel(d1, r1, 1e-4, "d_fmap1 (product form, 8 x 8 tiles)")
        assert_rel(d2, r2, 1e-4, "d_fmap2 (product form, 8 x 8 tiles)")


def test_backward_nchw_takes_the_product_form():
    """What an unmodified enet.py hands over (enet.py:203-216): a plain NCHW fp32 pyramid level, C = 128.  From 1024 edges on the
    backward reads a cached channels-last copy and returns fmap2_grad with fmap2's shape and channels-last strides."""
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(n=3, Np=30, C=128, H=24, W=32, E=1100, seed=23)
    g = torch.randn(1, 1100, 7, 7, 3, 3, generator=torch.Generator().manual_seed(2))
    r1, r2 = A.corr_backward(f1, f2, coords, ii, jj, g, R)
    f2d = f2.to(DEV)
    for rep in range(2):                                             # second call: the cached copy
        d1, d2 = cuda_corr.backward(f1.to(DEV), f2d, coords.to(DEV), ii.to(DEV), jj.to(DEV), g.to(DEV), R)
        assert cuda_corr.last_backward_path == _product_path()
        assert d2.shape == f2d.shape and d2.stride(2) == 1
        assert_rel(d1, r1, 1e-4, "d_fmap1 (NCHW, product form)")
        assert_rel(d2, r2, 1e-4, "d_fmap2 (NCHW, product form)")
    f2d[0, 0, 0, 0, 0] += 1.0                                        # an in-place write bumps the version: the copy is rebuilt
    f2[0, 0, 0, 0, 0] += 1.0
    r1, r2 = A.corr_backward(f1, f2, coords, ii, jj, g, R)
    d1, d2 = cuda_corr.backward(f1.to(DEV), f2d, coords.to(DEV), ii.to(DEV), jj.to(DEV), g.to(DEV), R)
    assert_rel(d1, r1, 1e-4, "d_fmap1 (NCHW, product form, after an in-place write)")
    # through autograd, like CorrLayer.backward: the gradient arrives at the NCHW leaf
    from devo_amd import altcorr
    leaf = f2.to(DEV).requires_grad_(True)
    out = altcorr.corr(f1.to(DEV), leaf, coords.to(DEV), ii.to(DEV), jj.to(DEV), R)
    out.backward(g.to(DEV))
    assert_rel(leaf.grad, r2, 1e-4, "d_fmap2 through autograd")


def test_autograd_layer_and_dropout():
    from devo_amd import altcorr
    f1, f2, coords, ii, jj, R = _case(n=3, Np=12, C=32, H=20, W=24, E=40, seed=23)
    a = f1.to(DEV).requires_grad_(True)
    b = channels_last5(f2.to(DEV)).requires_grad_(True)
    out = altcorr.corr(a, b, coords.to(DEV), ii.to(DEV), jj.to(DEV), R, 1)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(2))
    out.backward(g.to(DEV))
    r1, r2 = A.corr_backward(f1, f2, coords, ii, jj, g, R)
    assert_rel(a.grad, r1, 1e-4, "autograd d1")
    assert_rel(b.grad, r2, 1e-4, "autograd d2")
    # dropout < 1 keeps a random subset of edges (correlation.py:20-25): gradient magnitude shrinks, stays finite
    a.grad = None
    torch.manual_seed(0)
    altcorr.corr(a, b, coords.to(DEV), ii.to(DEV), jj.to(DEV), R, 0.2).backward(g.to(DEV))
    assert torch.isfinite(a.grad).all() and a.grad.abs().sum() < r1.abs().sum()


def test_patchify():
    from devo_amd.backends import cuda_corr
    from devo_amd import altcorr
    g = torch.Generator().manual_seed(29)
    net = torch.randn(3, 16, 20, 24, generator=g)
    coords = torch.stack([torch.rand(3, 10, generator=g) * 30 - 3, torch.rand(3, 10, generator=g) * 26 - 3], -1)
    coords[:, :3] = coords[:, :3].floor()
    for R in (0, 1, 3):
        ref = A.patchify_forward(net, coords, R)
        got, = cuda_corr.patchify_forward(net.to(DEV), coords.to(DEV), R)
        assert torch.equal(got.cpu(), ref)                               # pure gather: bit-exact
        got_cl, = cuda_corr.patchify_forward(net.to(DEV).contiguous(memory_format=torch.channels_last), coords.to(DEV), R)
        assert torch.equal(got_cl.cpu(), ref)
        gr = torch.randn(ref.shape, generator=g)
        back, = cuda_corr.patchify_backward(net.to(DEV), coords.to(DEV), gr.to(DEV), R)
        assert_rel(back, A.patchify_backward(net, coords, gr, R), 1e-5, "patchify bwd")
    x = net.to(DEV).requires_grad_(True)
    ci = coords.floor().to(DEV)
    y = altcorr.patchify(x, ci, 1)
    assert torch.equal(y.detach().cpu(), A.patchify(net, coords.floor(), 1))
    y.sum().backward()
    assert torch.isfinite(x.grad).all()
    assert torch.equal(cuda_corr.patchify_forward(net.to(DEV).half(), coords.to(DEV), 1)[0].cpu(), A.patchify_forward(net.half(), coords, 1))


def test_batch_of_two_and_plan_independence():
    """B = 2 (the reference kernels index coords[n][m]...), and: the locality plan only reorders work."""
    from devo_amd.backends import cuda_corr
    g = torch.Generator().manual_seed(31)
    B, E, Np, n, C, H, W, R = 2, 300, 10, 3, 32, 24, 32, 3
    f1 = torch.randn(B, Np, C, 3, 3, generator=g)
    f2 = torch.randn(B, n, C, H, W, generator=g)
    coords = torch.rand(B, E, 2, 3, 3, generator=g) * torch.tensor([W + 6.0, H + 6.0]).view(1, 1, 2, 1, 1) - 3.0
    ii = torch.randint(0, Np, (E,), generator=g)
    jj = torch.randint(0, n, (E,), generator=g)
    ref = A.corr_forward(f1, f2, coords, ii, jj, R)
    f2d = channels_last5(f2.to(DEV))
    args = (f1.to(DEV), f2d, coords.to(DEV), ii.to(DEV), jj.to(DEV))
    out_auto, = cuda_corr.forward(*args, R)                       # B*E < PLAN_MIN_EDGES: list order
    assert_rel(out_auto, ref, 1e-4, "B=2")
    plan = cuda_corr.plan(args[2], args[4], n, H)
    assert sorted(plan[:B * E].cpu().tolist()) == list(range(B * E))  # a permutation of the edge slots (+ heavy count)
    assert 0 <= int(plan[B * E]) <= B * E
    out_plan = torch.empty_like(out_auto)
    cuda_corr.forward_into(out_plan, *args, R, 7 * 7 * 9, 1, 0, order=plan)
    assert torch.equal(out_plan, out_auto)                        # bit-identical with and without the plan


def test_rejects_bad_arguments():
    from devo_amd.backends import cuda_corr
    f1 = torch.zeros(1, 2, 16, 3, 3, device=DEV)
    f2 = torch.zeros(1, 2, 16, 8, 8, device=DEV)
    c = torch.zeros(1, 4, 2, 3, 3, device=DEV)
    i = torch.zeros(4, dtype=torch.long, device=DEV)
    with pytest.raises(RuntimeError):
        cuda_corr.forward(f1, f2, c, i, i, 6)                     # radius > 5
    with pytest.raises(RuntimeError):
        cuda_corr.forward(f1, f2.half(), c, i, i, 3)              # dtype mismatch
    with pytest.raises(RuntimeError):
        cuda_corr.forward(f1, f2, torch.zeros(1, 4, 2, 2, 2, device=DEV), i, i, 3)   # patch size != 3
    out, = cuda_corr.forward(f1, f2, c[:, :0], i[:0], i[:0], 3)   # empty edge list
    assert out.shape == (1, 0, 7, 7, 3, 3)


def test_fused_pyramid_single_launch_matches_per_level_launches():
    """devo_corr_forward_pyramid2 (both levels in one launch) == one devo_corr_forward per level, bit for bit, for
    channels-last and channel-blocked pyramids, with and without a plan; NCHW falls back to per-level launches."""
    from devo_amd import altcorr
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(H=32, W=48, E=300, seed=23, spread=1.6)
    f2b = torch.nn.functional.avg_pool2d(f2[0], 4, 4)[None]
    args = (coords.to(DEV), ii.to(DEV), jj.to(DEV))
    for lay in (channels_last5, lambda t: altcorr.channel_blocked(t, 8), lambda t: t):
        pyr = [lay(f2.to(DEV)), lay(f2b.to(DEV))]
        per = torch.empty(1, len(ii), 2 * 49 * 9, device=DEV)
        for lvl, s in enumerate((1, 4)):
            cuda_corr.forward_into(per, f1.to(DEV), pyr[lvl], args[0], args[1], args[2], R, 2 * 49 * 9, 2, lvl, coord_div=float(s))
        fused = cuda_corr.forward_pyramid(f1.to(DEV), pyr, *args, R, (1, 4))
        assert torch.equal(fused, per)
        order = cuda_corr.plan(args[0], args[2], f2.shape[1], f2.shape[3])
        assert torch.equal(cuda_corr.forward_pyramid(f1.to(DEV), pyr, *args, R, (1, 4), order=order), per)


@pytest.mark.parametrize("R", [0, 1, 2, 4])
def test_forward_other_radii(R):
    """every radius the reference accepts below r = 5 (correlation.py callers use 3; patchify uses 0/1)"""
    c = _case(R=R, seed=40 + R, E=120)
    ref = A.corr_forward(*c)
    for layout in ("cl", "blk8", "nchw"):
        assert_rel(_run(*c, layout=layout), ref, 1e-4, f"corr fwd {layout} R={R}")


def test_coord_div_equals_divided_coordinates_everywhere():
    """forward_into(coords, coord_div=s) == forward_into(coords / s) bit for bit in the staged AND the generic kernel.
    The kernel performs a correctly rounded IEEE division: for DEVO's power-of-two scales that is the same number
    however the caller divides; for other scales it equals a true division (tensor divisor) — torch's CUDA/HIP
    `tensor / python_scalar` multiplies by the reciprocal instead and can differ in the last bit."""
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(E=150, seed=51)
    args = (f1.to(DEV), ii.to(DEV), jj.to(DEV))
    c = coords.to(DEV)
    for lay in (channels_last5, lambda t: t):
        fm = lay(f2.to(DEV))
        for s in (4.0, 3.0):
            a = torch.empty(1, len(ii), 49 * 9, device=DEV); b = torch.empty_like(a)
            cuda_corr.forward_into(a, args[0], fm, c, args[1], args[2], R, 49 * 9, 1, 0, coord_div=s)
            cuda_corr.forward_into(b, args[0], fm, c / torch.full((), s, device=DEV), args[1], args[2], R, 49 * 9, 1, 0)
            assert torch.equal(a, b)
            if s == 4.0:
                cuda_corr.forward_into(b, args[0], fm, c / s, args[1], args[2], R, 49 * 9, 1, 0)
                assert torch.equal(a, b)


def test_fused_pyramid_odd_edge_count_and_batch_of_two():
    """E not a multiple of 8 (the fused launch pads every level to whole groups of 8 workgroups) and B = 2"""
    from devo_amd.backends import cuda_corr
    B, n, Np, C, H, W, E, R = 2, 3, 10, 128, 32, 48, 2053, 3
    g = torch.Generator().manual_seed(61)
    f1 = (torch.randn(B, Np, C, 3, 3, generator=g) / 4).to(DEV)
    f2 = torch.randn(B, n, C, H, W, generator=g) / 4
    f2b = torch.stack([torch.nn.functional.avg_pool2d(f2[b], 4, 4) for b in range(B)])
    coords = (torch.rand(B, E, 2, 3, 3, generator=g) * torch.tensor([W, H]).view(1, 1, 2, 1, 1)).to(DEV)
    ii = torch.randint(0, Np, (E,), generator=g).to(DEV)
    jj = torch.randint(0, n, (E,), generator=g).to(DEV)
    pyr = [channels_last5(f2.to(DEV)), channels_last5(f2b.to(DEV))]
    fused = cuda_corr.forward_pyramid(f1, pyr, coords, ii, jj, R, (1, 4))          # B*E >= PLAN_MIN_EDGES: planned
    per = torch.empty_like(fused)
    for lvl, s in enumerate((1, 4)):
        cuda_corr.forward_into(per, f1, pyr[lvl], coords, ii, jj, R, 2 * 49 * 9, 2, lvl, coord_div=float(s))
    assert torch.equal(fused, per)
    ref = torch.stack([A.corr_forward(f1.cpu(), f2, coords.cpu(), ii.cpu(), jj.cpu(), R),
                       A.corr_forward(f1.cpu(), f2b, coords.cpu() / 4, ii.cpu(), jj.cpu(), R)], -1)
    assert_rel(fused, ref.view(B, E, -1), 1e-4, "fused pyramid B=2")


def test_build_pyramid_blocked():
    """devo.py:526-527 / utils.py:70-79 in one kernel: blocked level 0 is a bit-exact re-layout, level 1 the 4x4 mean
    (torch's avg_pool2d sums in another order: 1e-6), odd sizes floor like avg_pool2d, ring-buffer slots, fp16."""
    from devo_amd import altcorr
    g = torch.Generator().manual_seed(71)
    for (n, C, H, W), dt in (((3, 128, 30, 44), torch.float32), ((2, 16, 17, 150), torch.float32), ((2, 64, 32, 48), torch.float16)):
        f = (torch.randn(1, n, C, H, W, generator=g) / 4).to(DEV, dt)
        l0, l1 = altcorr.build_pyramid(f)
        assert torch.equal(l0, altcorr.channel_blocked(f, 8))
        ref1 = torch.nn.functional.avg_pool2d(f[0].float(), 4, 4)[None]
        assert l1.shape == (1, n, C // 8, H // 4, W // 4, 8)
        got1 = l1.permute(0, 1, 2, 5, 3, 4).reshape(1, n, C, H // 4, W // 4).float()
        assert_rel(got1, ref1, 1e-6 if dt == torch.float32 else 2e-3, "pooled level")
    ring0 = torch.zeros(1, 5, 16, 30, 44, 8, device=DEV); ring1 = torch.zeros(1, 5, 16, 7, 11, 8, device=DEV)
    f = torch.randn(1, 1, 128, 30, 44, generator=g).to(DEV)
    altcorr.build_pyramid(f, out=(ring0, ring1), slot=3)
    a0, a1 = altcorr.build_pyramid(f)
    assert torch.equal(ring0[:, 3:4], a0) and torch.equal(ring1[:, 3:4], a1) and float(ring0[:, :3].abs().max()) == 0.0


@pytest.mark.parametrize("seed", range(8))
def test_forward_random_configurations(seed):
    """random small problems: channel counts incl. non-multiples of 8 (generic kernel), tiny images (windows larger than
    the image), every radius, few / many edges, all layouts the configuration allows — always against the oracle"""
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    C = [8, 16, 24, 64, 128, 12][ri(0, 5)]
    R = ri(0, 5)
    H, W = ri(4, 40), ri(4, 48)
    c = _case(n=ri(1, 4), Np=ri(1, 9), C=C, H=H, W=W, E=ri(3, 260), R=R, seed=2000 + seed, spread=[0.3, 1.0, 2.5][ri(0, 2)])
    ref = A.corr_forward(*c)
    layouts = ["cl", "nchw"] + (["blk8"] if C % 8 == 0 else [])
    for layout in layouts:
        assert_rel(_run(*c, layout=layout), ref, 1e-4, f"seed {seed} C={C} R={R} {H}x{W} {layout}")


def test_plan_started_by_the_reprojection_kernel_equals_plan():
    """cuda_ba.transform(..., plan_for=...) + cuda_corr.plan_finish == cuda_corr.plan on the same coordinates: the same
    heavy set in front and the same multiset of edges in every (frame, band) bin (order inside a bin is free)"""
    from devo_amd import synth
    from devo_amd.backends import cuda_ba, cuda_corr
    n, M, H, W = 6, 40, 60, 80
    poses, (patches, _), intr = synth.make_poses(n, 9), synth.make_patches(n, M, H, W, seed=9), synth.make_intrinsics(n, H, W)
    ii, jj, kk = (t.to(DEV) for t in synth.full_graph(n, M))
    args = (poses.to(DEV), patches.to(DEV), intr.to(DEV), ii, jj, kk)
    coords, buf = cuda_ba.transform(*args, layout="2pp", plan_for=(n, H, 3))
    assert torch.equal(coords, cuda_ba.transform(*args, layout="2pp"))
    a = cuda_corr.plan_finish(buf, jj, n, H, 3).cpu()
    b = cuda_corr.plan(coords, jj, n, H).cpu()
    E = ii.numel()
    assert sorted(a[:E].tolist()) == list(range(E)) and int(a[E]) == int(b[E])
    nh = int(a[E])
    assert sorted(a[:nh].tolist()) == sorted(b[:nh].tolist())
    key = lambda o: torch.stack([jj.cpu()[o[nh:E].long()], (coords.cpu()[0, o[nh:E].long(), 1, 1, 1].clamp(0, H - 1) / 16).floor().long()], 1)
    assert torch.equal(key(a), key(b))                              # same (frame, band) sequence after the heavy list


@pytest.mark.parametrize("n,M,H,W", [(15, 96, 120, 160), (12, 300, 96, 128)])
def test_plan_by_several_workgroups_is_a_sorted_permutation(n, M, H, W):
    """above 2048 edges the ordering step runs as several independent workgroups (corr_plan.h), each owning a range of bins: the
    result is a permutation, heavy list first, target frames ascending behind it, every frame's edges complete"""
    from devo_amd import synth
    from devo_amd.backends import cuda_ba, cuda_corr
    poses, (patches, _), intr = synth.make_poses(n, 5), synth.make_patches(n, M, H, W, seed=5), synth.make_intrinsics(n, H, W)
    ii, jj, kk = (t.to(DEV) for t in synth.full_graph(n, M))
    coords = cuda_ba.transform(poses.to(DEV), patches.to(DEV), intr.to(DEV), ii, jj, kk, layout="2pp")
    E = ii.numel()
    for radius in (3, 5):
        o = cuda_corr.plan(coords, jj, n, H, radius=radius).cpu()
        nh = int(o[E])
        assert 0 <= nh <= E and sorted(o[:E].tolist()) == list(range(E))
        fr = jj.cpu()[o[nh:E].long()]
        assert bool((fr[1:] >= fr[:-1]).all())
        band = (coords.cpu()[0, o[nh:E].long(), 1, 1, 1].clamp(0, H - 1) / 64).floor().long()      # blocks of 4 bands ascend inside a frame
        key = fr * 64 + band
        assert bool((key[1:] >= key[:-1]).all())
    # spread-out patches (boxes the tile cannot hold) form the heavy list in front, filled by the first workgroup only; the
    # lookup with this plan is bit-identical to the lookup in list order
    g = torch.Generator().manual_seed(3)
    spread = torch.rand(E, generator=g) < 0.03
    c2 = coords.clone()
    ctr = c2[0, :, :, 1:2, 1:2]
    c2[0, spread.to(DEV)] = (ctr + 9.0 * (c2[0] - ctr))[spread.to(DEV)]
    o = cuda_corr.plan(c2, jj, n, H, radius=3).cpu()
    nh = int(o[E])
    assert sorted(o[:E].tolist()) == list(range(E))
    heavy = torch.zeros(E, dtype=torch.bool)
    heavy[o[:nh].long()] = True
    assert bool(heavy[spread].all())                               # every spread patch is in the heavy list
    C = 32
    f1 = torch.randn(1, n * M, C, 3, 3, generator=g).to(DEV)
    f2 = channels_last5(torch.randn(1, n, C, H // 4, W // 4, generator=g).to(DEV))
    c4 = (c2 / 4).contiguous()
    plan4 = cuda_corr.plan(c4, jj, n, H // 4, radius=3)
    ident = torch.cat([torch.arange(E, dtype=torch.int32), torch.zeros(1, dtype=torch.int32)]).to(DEV)      # list order, no heavy edges
    out_list, out_plan = torch.empty(1, E, 7, 7, 3, 3, device=DEV), torch.empty(1, E, 7, 7, 3, 3, device=DEV)
    cuda_corr.forward_into(out_list, f1, f2, c4, kk, jj, 3, 7 * 7 * 9, 1, 0, order=ident)
    cuda_corr.forward_into(out_plan, f1, f2, c4, kk, jj, 3, 7 * 7 * 9, 1, 0, order=plan4)
    assert torch.equal(out_plan, out_list)


@pytest.mark.parametrize("which", ["staged kernel", "segment-reduced backward", "atomic backward"])
def test_other_kernels_stay_covered(which):
    """fp32 / fp16 lookups with C = 128 take the per-edge matrix-core kernel by default.  DEVO_CORR_MFMA=0 (read once per process)
    routes them through the staged tap-centric kernel, DEVO_CORR_BWD_SEG=1 the backward through the tile kernel, DEVO_CORR_BWD_ATOMIC=1
    the channels-last C % 128 == 0 backward (product form by default) through the one-kernel atomic path: same parity tests.
    DEVO_CORR_MM=0 routes fused and per-level lookups through the 4x4 matrix-core kernel (exact fp32 products) instead of the dense-product one."""
    if os.environ.get("DEVO_CORR_MFMA", "1") == "0" or os.environ.get("DEVO_CORR_BWD_SEG") or os.environ.get("DEVO_CORR_BWD_ATOMIC"):
        pytest.skip("already running on another kernel")
    env = dict(os.environ)
    sel = "test_forward_fp32 or test_forward_wide_spread or test_channel_blocked or test_fused_pyramid or test_batch_of_two or test_forward_other_radii or test_coord_div"
    if which == "staged kernel":
        env["DEVO_CORR_MFMA"] = "0"
    elif which == "atomic backward":
        env["DEVO_CORR_BWD_ATOMIC"] = "1"
        sel = "test_backward_product_form"
    else:                                                           # opt-in: d_fmap2 tile by tile in LDS instead of global atomics
        env["DEVO_CORR_BWD_SEG"] = "1"
        sel = "test_backward_fp32 or test_autograd_layer or test_dtype_coverage or test_backward_product_form"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", sel],
                       env=env, capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("R,spread,scales", [(1, 1.0, (1, 4)), (5, 1.0, (1, 4)), (3, 3.5, (1, 4)), (3, 9.0, (1, 4)),
                                              (3, 1.0, (1.0, 3.0)), (2, 2.0, (2.0, 8.0))])
def test_fused_pyramid_radii_spreads_and_scales(R, spread, scales):
    """the wave that does both levels of an edge (fp32, C = 128): other radii (raw-window and box layouts of the result
    area), boxes of 3+ passes and window-by-window passes (spread patches), and level scales that are not powers of two
    (true division instead of the exact reciprocal) — bit-identical to one launch per level, within 1e-4 of the oracle"""
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, _ = _case(H=32, W=48, E=257, R=R, seed=100 + R, spread=spread)
    f2b = torch.nn.functional.avg_pool2d(f2[0], 4, 4)[None]
    pyr = [channels_last5(f2.to(DEV)), channels_last5(f2b.to(DEV))]
    args = (coords.to(DEV), ii.to(DEV), jj.to(DEV))
    Dm = 2 * R + 1
    per = torch.empty(1, len(ii), 2 * Dm * Dm * 9, device=DEV)
    for lvl, s in enumerate(scales):
        cuda_corr.forward_into(per, f1.to(DEV), pyr[lvl], *args, R, 2 * Dm * Dm * 9, 2, lvl, coord_div=float(s))
    fused = cuda_corr.forward_pyramid(f1.to(DEV), pyr, *args, R, scales)
    assert torch.equal(fused, per)
    div = lambda s: coords / torch.tensor(float(s))                   # true division (a python scalar would multiply)
    ref = torch.stack([A.corr_forward(f1, f2, div(scales[0]), ii, jj, R), A.corr_forward(f1, f2b, div(scales[1]), ii, jj, R)], -1)
    assert_rel(fused, ref.view(1, len(ii), -1), 1e-4, f"fused R={R} spread={spread} scales={scales}")


def test_fused_pyramid_fp16_storage():
    """fp16 features on the matrix cores (4 channels per MFMA, exact products, fp32 accumulation): the two-level wave equals
    the per-level launches bit for bit and the fp32 oracle within the fp16-storage tolerance (2e-3)"""
    from devo_amd import altcorr
    from devo_amd.backends import cuda_corr
    for R, spread, lay in ((3, 1.0, "cl"), (3, 3.5, "blk8"), (5, 1.0, "blk8"), (3, 9.0, "cl")):
        f1, f2, coords, ii, jj, _ = _case(H=32, W=48, E=257, R=R, seed=200 + R, spread=spread)
        f2b = torch.nn.functional.avg_pool2d(f2[0], 4, 4)[None]
        conv = (lambda t: channels_last5(t)) if lay == "cl" else (lambda t: altcorr.channel_blocked(t, 8))
        pyr = [conv(f2.to(DEV).half()), conv(f2b.to(DEV).half())]
        args = (coords.to(DEV), ii.to(DEV), jj.to(DEV))
        Dm = 2 * R + 1
        per = torch.empty(1, len(ii), 2 * Dm * Dm * 9, device=DEV, dtype=torch.float16)
        for lvl, s_ in enumerate((1, 4)):
            cuda_corr.forward_into(per, f1.to(DEV).half(), pyr[lvl], *args, R, 2 * Dm * Dm * 9, 2, lvl, coord_div=float(s_))
        fused = cuda_corr.forward_pyramid(f1.to(DEV).half(), pyr, *args, R, (1, 4))
        assert torch.equal(fused, per)
        ref = torch.stack([A.corr_forward(f1.half().float(), f2.half().float(), coords, ii, jj, R),
                           A.corr_forward(f1.half().float(), f2b.half().float(), coords / 4, ii, jj, R)], -1)
        assert_rel(fused.float(), ref.view(1, len(ii), -1), 2e-3, f"fused fp16 R={R} spread={spread} {lay}")


@pytest.mark.parametrize("C,dtype", [(64, torch.float32), (256, torch.float16)])
def test_matrix_core_kernel_other_channel_counts(C, dtype):
    """4 steps per pass (fp32 C = 64) and 8 (fp16 C = 256) of the matrix-core kernel, per level and fused"""
    from devo_amd import altcorr
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(C=C, H=32, W=48, E=150, seed=300 + C, spread=1.5)
    tol = 1e-4 if dtype == torch.float32 else 2e-3
    q = (lambda t: t) if dtype == torch.float32 else (lambda t: t.half().float())
    ref0 = A.corr_forward(q(f1), q(f2), coords, ii, jj, R)
    for lay in ("cl", "blk8"):
        assert_rel(_run(f1, f2, coords, ii, jj, R, layout=lay, dtype=dtype).float(), ref0, tol, f"C={C} {lay}")
    f2b = torch.nn.functional.avg_pool2d(f2[0], 4, 4)[None]
    pyr = [altcorr.channel_blocked(f2.to(DEV, dtype), 8), altcorr.channel_blocked(f2b.to(DEV, dtype), 8)]
    fused = cuda_corr.forward_pyramid(f1.to(DEV, dtype), pyr, coords.to(DEV), ii.to(DEV), jj.to(DEV), R, (1, 4))
    ref = torch.stack([ref0, A.corr_forward(q(f1), q(f2b), coords / 4, ii, jj, R)], -1)
    assert_rel(fused.float(), ref.view(1, len(ii), -1), tol, f"fused C={C}")


def test_differentiable_fused_pyramid_lookup_matches_the_two_level_composition():
    """altcorr.corr_pyramid with gradients (CorrPyramidLayer: one fused forward launch, per-level backward kernels on the strided halves
    of the gradient) against torch.stack([corr(level 0), corr(level 1 at coords / 4)], -1) — values and both feature gradients"""
    from devo_amd import altcorr
    torch.manual_seed(5)
    n, C, H, W, E, R = 4, 128, 48, 64, 1500, 3
    f0 = torch.randn(1, n, C, H, W, device=DEV) * 0.3
    pyr = [altcorr.channels_last(f0).requires_grad_(True), altcorr.channels_last(torch.nn.functional.avg_pool2d(f0[0], 4, 4)[None]).requires_grad_(True)]
    g1 = (torch.randn(1, 60, C, 3, 3, device=DEV) * 0.3).requires_grad_(True)
    ii = torch.randint(0, 60, (E,), device=DEV)
    jj = torch.randint(0, n, (E,), device=DEV)
    coords = torch.stack([torch.rand(1, E, 3, 3, device=DEV) * (W + 8) - 4, torch.rand(1, E, 3, 3, device=DEV) * (H + 8) - 4], 2)
    gout = torch.randn(1, E, 2 * 49 * 9, device=DEV)
    outs = []
    for fused in (True, False):
        for t in pyr + [g1]:
            t.grad = None
        if fused:
            y = altcorr.corr_pyramid(g1, pyr, coords, ii, jj, R, (1, 4))
        else:
            y = torch.stack([altcorr.corr(g1, pyr[0], coords / 1, ii, jj, R), altcorr.corr(g1, pyr[1], coords / 4, ii, jj, R)], -1).view(1, E, -1)
        y.backward(gout)
        outs.append([y.detach().clone(), g1.grad.clone(), pyr[0].grad.clone(), pyr[1].grad.clone()])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-6) + 1e-6


@pytest.mark.parametrize("case", ["x4096", "x1/4096", "outlier_1e5", "fp16_denormal_heavy", "fp32_denormals", "frames_of_different_scale"])
@pytest.mark.parametrize("layout", ["blk8", "nchw"])
def test_fp32_lookup_at_any_feature_magnitude(case, layout):
    """The dense-product kernel multiplies fp32 features as fp16 hi + lo pairs (csrc/corr_mm.h).  The pairs are formed from the value
    scaled by a power of two per patch / per frame (devo_corr_patch_transpose, devo_corr_pyramid_split), so no magnitude overflows
    fp16 (the reference's fp32 kernel returns finite numbers there, correlation_kernel.cu:82-136) or falls below its normal range:
    the fp32 tolerance (1e-4 of the output scale, against the fp64 oracle) holds for features of any scale, and nothing is inf."""
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(E=1200, Np=40, seed=41)
    if case == "x4096":
        f1, f2 = f1 * 4096.0, f2 * 4096.0                        # |x| up to ~5e3: products up to 2.5e7, a plain fp16 hi would reach 65504 at x16 more
        f2[0, 1] *= 64.0                                         # ... and one frame beyond fp16's range altogether (|x| up to 3e5)
    elif case == "x1/4096":
        f1, f2 = f1 / 4096.0, f2 / 4096.0                        # |x| ~ 6e-5: at fp16's normal / denormal boundary
    elif case == "outlier_1e5":
        f2[0, 2, 17, 11, 13] = 1.0e5                             # one value beyond fp16's largest (65504)
        f1[0, 3, 5, 1, 1] = -2.0e5
    elif case == "fp16_denormal_heavy":
        g = torch.Generator().manual_seed(5)
        f2 = f2 * torch.where(torch.rand(f2.shape, generator=g) < 0.7, 1.0e-6, 1.0)      # 70 % of the map far below fp16's normal range
        f1 = f1 * 1.0e-6                                         # a patch operand that is below it entirely
    elif case == "fp32_denormals":
        g = torch.Generator().manual_seed(6)
        f2 = f2 * torch.where(torch.rand(f2.shape, generator=g) < 0.5, 1.0e-39, 1.0)     # fp32 denormals among O(1) values
    elif case == "frames_of_different_scale":
        for k in range(f2.shape[1]):
            f2[0, k] *= 10.0 ** (3 * k - 4)                      # 1e-4 .. 1e5: per-frame exponents
    was = cuda_corr.MM_KERNEL
    try:
        cuda_corr.MM_KERNEL = True
        got = _run(f1, f2, coords, ii, jj, R, layout=layout)
    finally:
        cuda_corr.MM_KERNEL = was
    assert torch.isfinite(got).all(), "non-finite output"
    ref = A.corr_forward(f1.double(), f2.double(), coords, ii, jj, R)
    if case == "frames_of_different_scale":                      # every frame against its own output scale
        for k in range(f2.shape[1]):
            sel = jj == k
            assert_rel(got[:, sel.to(DEV)], ref[:, sel], 1e-4, f"frame {k}")
    else:
        assert_rel(got, ref, 1e-4, case)


def _group_case(n=5, Np=60, C=128, H=48, W=64, E=6000, seed=0, heavy_frac=0.03):
    """Compact patches (pixel spacing 0.6 - 1.5 px) all over and around an H x W frame, a few widely spread ones, a few far outside."""
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(1, Np, C, 3, 3, generator=g) / 4
    f2 = torch.randn(1, n, C, H, W, generator=g) / 4
    base = torch.stack([torch.rand(E, generator=g) * (W + 40) - 20, torch.rand(E, generator=g) * (H + 40) - 20], 1)
    oy, ox = torch.meshgrid(torch.arange(3.) - 1, torch.arange(3.) - 1, indexing="ij")
    off = torch.stack([ox, oy], 0)
    scale = 0.6 + 0.9 * torch.rand(E, 1, 1, 1, generator=g)
    wide = torch.rand(E, generator=g) < heavy_frac
    scale[wide] = scale[wide] * 6.0                                    # level-1 boxes that leave their group's region / level-0 boxes beyond 128 positions
    coords = base[:, :, None, None] + scale * off[None] + 0.2 * torch.randn(E, 2, 3, 3, generator=g)
    coords[:7] = -300.0 + off                                          # dead edges
    ii = torch.randint(0, Np, (E,), generator=g)
    jj = torch.randint(0, n, (E,), generator=g)
    return f1, f2, coords[None].contiguous(), ii, jj


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("H,W", [(48, 64), (44, 52), (120, 160)])
def test_group_form_reads_level_1_from_lds_and_changes_nothing(dtype, H, W):
    """The fused two-level lookup with a GROUP plan (cuda_corr.plan(..., width, l1 = 4); csrc/corr_mm.h, NW > 1): workgroups stage their
    group's 15 x 15 region of level 1 in LDS and their waves read level 1's tiles from there.  Same products in the same order: the
    output is bit-identical to the per-edge form's (an edge plan), for every class of edge (compact, heavy, dead, frame borders, frames
    whose size is not a multiple of the group tile) — and within the fp32 / fp16 tolerances of the oracle."""
    from devo_amd import altcorr, synth
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj = _group_case(H=H, W=W, seed=H + W, E=6000 if H < 100 else 9000)
    n, E = f2.shape[1], coords.shape[1]
    d = lambda t: t.to(DEV)
    f2d = d(f2).to(dtype)
    pyr = [altcorr.channel_blocked(f2d, 8), altcorr.channel_blocked(synth.pyramid_l1(f2d.float()).to(dtype), 8)]
    g = d(f1).to(dtype)
    out = {}
    for kind in ("edges", "groups"):
        order = cuda_corr.plan(d(coords), d(jj), n, H, 1.0, 3, width=W if kind == "groups" else 0, l1=4 if kind == "groups" else 0)
        assert cuda_corr.plan_kind(order, E) == (1 if kind == "groups" else 0)
        res = torch.full((1, E, 49 * 18), float("nan"), dtype=dtype, device=DEV)
        cuda_corr.forward_pyramid(g, pyr, d(coords), d(ii), d(jj), 3, (1, 4), out=res, order=order)
        out[kind] = res
    assert torch.equal(out["groups"], out["edges"])
    # the group plan: a permutation, heavy first, dead last, every group's slots contiguous and inside one frame
    order = cuda_corr.plan(d(coords), d(jj), n, H, 1.0, 3, width=W, l1=4).cpu()
    perm = order[:E]
    assert sorted(perm.tolist()) == list(range(E))
    nh, nd = int(order[E]), int(order[2 * E + 1])
    gy, gx = (H // 4 + 5) // 6, (W // 4 + 5) // 6
    nb = n * gy * gx + 1
    starts = order[2 * E + 2: 2 * E + 2 + nb + 1]
    assert int(starts[0]) == nh and int(starts[nb]) == E and int(starts[nb - 1]) == E - nd and nd >= 7
    assert bool((starts[1:] >= starts[:-1]).all())
    fr = jj[perm[nh:E - nd]]
    own = torch.repeat_interleave(torch.arange(nb - 1) // (gy * gx), (starts[1:nb] - starts[:nb - 1]).long())
    assert torch.equal(fr, own)
    assert nh > 0 and nh < E // 2                                   # (pixel spacings up to 1.5 px: many level-0 boxes beyond 128 positions)
    # oracle on a sample
    sel = torch.randperm(E, generator=torch.Generator().manual_seed(1))[:400]
    q = lambda t: t.to(dtype).float()
    cs = coords[:, sel]
    r0 = A.corr_forward(q(f1), q(f2), cs, ii[sel], jj[sel], 3)
    r1 = A.corr_forward(q(f1), q(synth.pyramid_l1(f2d.float()).cpu()), cs / 4, ii[sel], jj[sel], 3)
    ref = torch.stack([r0, r1], -1).view(1, len(sel), -1)
    assert_rel(out["groups"][:, sel.to(DEV)].float(), ref, 1e-4 if dtype == torch.float32 else 2e-3, "group form vs oracle")


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_lookup_results_do_not_depend_on_the_plan(dtype):
    """A plan only decides which edges run together (and which of them the kernel meets first): with a plan made for OTHER coordinates — other
    bins, another heavy list — the fused two-level lookup and the per-level launches return the same bits.  This is what lets the second
    of DEVO's two per-level corr calls (devo.py:215-216) take the plan the first one made: two consecutive calls with the same index tensors
    return what two calls with their own plans return."""
    import bench
    from devo_amd import synth
    from devo_amd.backends import cuda_ba, cuda_corr
    cfg = synth.workload("cfg2")
    d, _ = bench.build_inputs(cfg, 1234, torch.device(DEV), dtype, "blk8")
    coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
    n, R, H = cfg["n"], cfg["R"], cfg["H"]
    E = d["ii"].numel()
    g = torch.Generator(device="cpu").manual_seed(0)
    other = coords + (torch.randn(1, E, 2, 1, 1, generator=g) * 40).to(DEV)
    far = coords.clone(); far[:, ::3] += 5000.0
    look = lambda order: cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4), order=order)
    own = cuda_corr.plan(coords, d["jj"], n, H, radius=R)
    out = look(own)
    for c2 in (other, far):
        foreign = cuda_corr.plan(c2, d["jj"], n, H, radius=R)
        assert not torch.equal(foreign[:E], own[:E]) and torch.equal(look(foreign), out)
    # the reference's per-level calls: the second takes the first one's plan (both bindings)
    ref = []
    for fm, s in zip(d["pyramid"], (1.0, 4.0)):
        cuda_corr._last_plan = None                                        # (every reference call makes its own plan)
        ref.append(cuda_corr._forward_ctypes(d["gmap"], fm, coords / s, d["kk"], d["jj"], R)[0])
    cuda_corr._last_plan = None
    jj2 = d["jj"].clone()                                                  # (fresh index tensor: own plans above, a handed-on plan below)
    for fwd in (cuda_corr.forward, cuda_corr._forward_ctypes):
        got = [fwd(d["gmap"], fm, coords / s, d["kk"], jj2, R)[0] for fm, s in zip(d["pyramid"], (1.0, 4.0))]
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])


def test_every_call_says_which_kernel_it_took():
    """cuda_corr.last_forward_path() / cuda_ba.last_path(): the kernel family of the calling thread's last call (devo_corr_forward_last_path,
    devo_ba_last_path) — a layout, dtype or size outside the fast kernels used to cost 2 - 24 x without a trace; the slow lookups also say
    so once on stderr (DEVO_LOG_FALLBACK=0: silent)."""
    import subprocess, sys, os, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, torch
        sys.path.insert(0, %r)
        import bench
        from devo_amd import synth
        from devo_amd.backends import cuda_ba, cuda_corr
        dev = torch.device("cuda", 0)
        cfg = synth.workload("cfg2")
        d, _ = bench.build_inputs(cfg, 1, dev, torch.float32, "blk8")
        coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
        n, R = cfg["n"], cfg["R"]
        cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4))
        print("A", cuda_corr.last_forward_path())
        nchw = d["pyramid"][0].permute(0, 1, 2, 5, 3, 4).reshape(1, n, 128, cfg["H"], cfg["W"]).contiguous()
        cuda_corr._forward_ctypes(d["gmap"].double(), nchw.double(), coords, d["kk"], d["jj"], R)
        print("B", cuda_corr.last_forward_path())
        ws = cuda_ba.workspace(d["ii"].numel(), d["patches0"].shape[1], n - 1, dev)
        tgt = coords[:, :, :, 1, 1].contiguous()
        cuda_ba.forward(d["poses0"].clone(), d["patches0"].clone(), d["intr"], tgt, d["weight"], d["lmbda"], d["ii"], d["jj"], d["kk"], 1, n, 2, ws=ws)
        print("C", cuda_ba.last_path())
    """ % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    out = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines() if line[:2] in ("A ", "B ", "C "))
    assert out.get("A") == "dense-product" and out.get("B") == "generic" and out.get("C") == "accumulate:register solve:chain", (r.stdout, r.stderr[-1500:])
    assert r.stderr.count("the generic (slowest") == 1                     # announced once
