"""GPU parity of the REGION-SHARED lookup kernel (devo_amd/csrc/corr_region.h) — the two-level pyramid lookup with a pyramid plan —
against the CPU oracle (correlation_kernel.cu:82-136,221-232): every storage layout it reads, fp32 / fp16, radii 0..5, channel counts,
batch 2, out-of-frame / integer / far-spread coordinates (the tap-by-tap path, the heavy list, the dead tail), rounds that cannot
share a region (random plan order), and bit-identity across plans that keep the classes.
Tolerance: 1e-4 relative to the output scale (fp32: fp16 hi + lo products, fp32 accumulation), 2e-3 for fp16 storage."""
import pytest
import torch
from oracle import altcorr as A
from util import assert_rel, channels_last5

import os
import subprocess
import sys
pytestmark = pytest.mark.gpu
DEV = "cuda"
REGION = os.environ.get("DEVO_CORR_REGION", "0") == "1"      # the library reads the switch once per process


def test_region_kernel_suite_in_a_sub_process():
    """the region-shared kernel is opt-in (DEVO_CORR_REGION=1, read once per process): this file runs again in a sub-process with it"""
    if REGION:
        pytest.skip("this is the sub-process")
    env = dict(os.environ); env["DEVO_CORR_REGION"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x"], env=env, capture_output=True, text=True,
                       timeout=1500, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def _case(n=3, Np=40, C=128, H=32, W=48, E=3000, R=3, seed=0, spread=1.0, B=1, far=0.03):
    """Edges clustered like DEVO's: patch centres all over (and a little beyond) the frame, 3x3 pixel grids with a random scale."""
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(B, Np, C, 3, 3, generator=g) / 4
    f2 = torch.randn(B, n, C, H, W, generator=g) / 4
    base = torch.stack([torch.rand(B, E, generator=g) * (W + 40) - 20, torch.rand(B, E, generator=g) * (H + 40) - 20], 2)
    oy, ox = torch.meshgrid(torch.arange(3.) - 1, torch.arange(3.) - 1, indexing="ij")
    off = torch.stack([ox, oy], 0)
    scale = spread * (0.6 + 0.9 * torch.rand(B, E, 1, 1, 1, generator=g))
    wide = torch.rand(B, E, 1, 1, 1, generator=g) < far                 # a few edges whose pixels lie far apart (heavy / tap by tap)
    scale = torch.where(wide, scale * 6.0, scale)
    coords = base[..., None, None] + scale * off + 0.2 * torch.randn(B, E, 2, 3, 3, generator=g)
    coords[:, 0] = coords[:, 0].round()                                  # exactly integer (dx = dy = 0)
    coords[:, 1] = -50.0 + off                                           # dead at both levels
    coords[:, 2] = torch.tensor([W + 2.0, H / 2.0])[:, None, None] + off  # level 0 misses the frame, level 1 touches it
    ii = torch.randint(0, Np, (E,), generator=g)
    jj = torch.randint(0, n, (E,), generator=g)
    return f1, f2, coords.contiguous(), ii, jj, R


def _oracle(f1, f2, coords, ii, jj, R):
    l1 = torch.nn.functional.avg_pool2d(f2.flatten(0, 1), 4, 4).view(*f2.shape[:3], f2.shape[3] // 4, f2.shape[4] // 4)
    c0 = A.corr_forward(f1, f2, coords, ii, jj, R)
    c1 = A.corr_forward(f1, l1, coords / 4, ii, jj, R)
    return torch.stack([c0, c1], -1).flatten(2), l1


def _layout(x, layout):
    from devo_amd import altcorr
    if layout == "cl":
        return channels_last5(x)
    return altcorr.channel_blocked(x, int(layout[3:]))


def _lookup(f1, f2, l1, coords, ii, jj, R, layout="blk8", dtype=torch.float32, order=None, want_plan=False):
    from devo_amd.backends import cuda_corr
    f1d, pyr = f1.to(DEV, dtype), [_layout(f2.to(DEV, dtype), layout), _layout(l1.to(DEV, dtype), layout)]
    cd, jd = coords.to(DEV), jj.to(DEV)
    if order is None:
        order = cuda_corr.plan(cd, jd, f2.shape[1], f2.shape[3], 1.0, R, width=f2.shape[4], l1=4)
    out = cuda_corr.forward_pyramid(f1d, pyr, cd, ii.to(DEV), jd, R, (1, 4), order=order)
    return (out, order) if want_plan else out


needs_region = pytest.mark.skipif(not REGION, reason="runs in the DEVO_CORR_REGION=1 sub-process")


@needs_region
@pytest.mark.parametrize("layout", ["blk8", "blk4", "blk16", "cl"])
def test_fp32_layouts(layout):
    c = _case(seed=1)
    ref, l1 = _oracle(*c)
    got, plan = _lookup(*c[:2], l1, *c[2:], layout=layout, want_plan=True)
    BE = c[2].shape[0] * c[2].shape[1]
    nh, nd = int(plan[BE]), int(plan[2 * BE + 1])
    assert 0 < nd < BE // 2 and 0 <= nh < BE // 4 and sorted(plan[:BE].cpu().tolist()) == list(range(BE))
    assert_rel(got, ref, 1e-4, f"region lookup fp32 {layout}")
    assert torch.count_nonzero(got[0, 1]) == 0                          # a dead edge is exactly zero


@needs_region
@pytest.mark.parametrize("layout", ["blk8", "blk16", "blk32", "cl"])
def test_fp16_layouts(layout):
    f1, f2, coords, ii, jj, R = _case(seed=2)
    f1, f2 = f1.half().float(), f2.half().float()
    ref, l1 = _oracle(f1, f2, coords, ii, jj, R)
    got = _lookup(f1, f2, l1.half().float(), coords, ii, jj, R, layout=layout, dtype=torch.float16)
    # level 1 of the oracle is pooled from the rounded level 0; the kernel reads a rounded copy of that
    ref1 = A.corr_forward(f1, l1.half().float(), coords / 4, ii, jj, R)
    ref = torch.stack([A.corr_forward(f1, f2, coords, ii, jj, R), ref1], -1).flatten(2)
    assert got.dtype == torch.float16
    assert_rel(got.float(), ref, 2e-3, f"region lookup fp16 {layout}")


@needs_region
@pytest.mark.parametrize("R", [0, 1, 2, 4, 5])
def test_radii(R):
    c = _case(seed=10 + R, R=R, E=2500, H=40, W=56)
    ref, l1 = _oracle(*c)
    assert_rel(_lookup(*c[:2], l1, *c[2:]), ref, 1e-4, f"region lookup R={R}")


@needs_region
@pytest.mark.parametrize("C,dtype", [(64, torch.float32), (32, torch.float32), (256, torch.float16), (64, torch.float16)])
def test_channel_counts(C, dtype):
    f1, f2, coords, ii, jj, R = _case(seed=20, C=C)
    if dtype == torch.float16:
        f1, f2 = f1.half().float(), f2.half().float()
    l1 = torch.nn.functional.avg_pool2d(f2.flatten(0, 1), 4, 4).view(*f2.shape[:3], f2.shape[3] // 4, f2.shape[4] // 4)
    if dtype == torch.float16:
        l1 = l1.half().float()
    ref = torch.stack([A.corr_forward(f1, f2, coords, ii, jj, R), A.corr_forward(f1, l1, coords / 4, ii, jj, R)], -1).flatten(2)
    got = _lookup(f1, f2, l1, coords, ii, jj, R, dtype=dtype)
    assert_rel(got.float(), ref, 1e-4 if dtype == torch.float32 else 2e-3, f"region lookup C={C}")


@needs_region
def test_batch_of_two():
    c = _case(seed=30, B=2, E=1800)
    ref, l1 = _oracle(*c)
    assert_rel(_lookup(*c[:2], l1, *c[2:]), ref, 1e-4, "region lookup B=2")


@needs_region
def test_far_spread_pixels_and_everything_outside():
    """many edges whose pixels lie far apart (the plan's heavy list -> per-edge kernel) and a frame nobody hits"""
    c = _case(seed=40, far=0.4, E=2200)
    ref, l1 = _oracle(*c)
    assert_rel(_lookup(*c[:2], l1, *c[2:]), ref, 1e-4, "heavy list")
    f1, f2, coords, ii, jj, R = _case(seed=41, E=2100)
    coords = coords + 500.0
    ref, l1 = _oracle(f1, f2, coords, ii, jj, R)
    got = _lookup(f1, f2, l1, coords, ii, jj, R)
    assert torch.count_nonzero(got) == 0 and torch.count_nonzero(ref) == 0


@needs_region
def test_a_plan_without_classes_takes_the_tap_by_tap_path_and_random_order_only_costs_time():
    """identity order, no heavy list, no dead tail: edges the rounds cannot take are computed tap by tap; unrelated neighbours in
    a chunk become rounds of their own.  Same results (tap-by-tap edges: plain fp32 sums instead of hi + lo products)."""
    c = _case(seed=50, far=0.1, E=2300)
    ref, l1 = _oracle(*c)
    E = c[2].shape[1]
    ident = torch.cat([torch.arange(E, dtype=torch.int32), torch.zeros(E + 2, dtype=torch.int32)]).to(DEV)
    assert_rel(_lookup(*c[:2], l1, *c[2:], order=ident), ref, 1e-4, "identity plan")


@needs_region
def test_plans_that_keep_the_classes_do_not_change_one_bit():
    """the plan decides which edges share a round, never a result: shuffle the heavy slots, the live slots and the dead slots among
    themselves"""
    c = _case(seed=60, E=4000)
    _, l1 = _oracle(*c)
    out, plan = _lookup(*c[:2], l1, *c[2:], want_plan=True)
    E = c[2].shape[1]
    nh, nd = int(plan[E]), int(plan[2 * E + 1])
    g = torch.Generator().manual_seed(0)
    p = plan.clone().cpu()
    for lo, hi in ((0, nh), (nh, E - nd), (E - nd, E)):
        if hi > lo:
            p[lo:hi] = p[lo:hi][torch.randperm(hi - lo, generator=g)]
    assert torch.equal(_lookup(*c[:2], l1, *c[2:], order=p.to(DEV)), out)


@needs_region
def test_drop_in_path_builds_its_own_plan_and_matches_per_level_calls():
    """forward_pyramid without a plan (>= 2048 edges: it builds the pyramid plan itself) against two single-level lookups of the
    per-edge kernel"""
    from devo_amd.backends import cuda_corr
    f1, f2, coords, ii, jj, R = _case(seed=70)
    l1 = torch.nn.functional.avg_pool2d(f2.flatten(0, 1), 4, 4).view(*f2.shape[:3], f2.shape[3] // 4, f2.shape[4] // 4)
    d = lambda t: t.to(DEV)
    pyr = [_layout(d(f2), "blk8"), _layout(d(l1), "blk8")]
    fused = cuda_corr.forward_pyramid(d(f1), pyr, d(coords), d(ii), d(jj), R, (1, 4))
    c0, = cuda_corr.forward(d(f1), pyr[0], d(coords), d(ii), d(jj), R)
    c1, = cuda_corr.forward(d(f1), pyr[1], d(coords) / 4, d(ii), d(jj), R)
    assert_rel(fused, torch.stack([c0, c1], -1).flatten(2), 2e-5, "region kernel vs per-edge kernel")
