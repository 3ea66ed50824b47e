"""GPU parity of the event voxelisation / standardisation (devo_amd/events.py + csrc/events.hip through
the C ABI) against the reference-generated golden (tests/golden/events_f32.npz) and the CPU oracle (oracle/events.py).
The voting order differs from the reference's eight sequential index_add_ passes, so the comparison is to fp32 rounding
(1e-5 of the grid's scale), not bit for bit."""
import os
import numpy as np
import pytest
import torch
from oracle import events as EV
from util import assert_rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_voxel_grid_and_std_match_reference_golden(golden_dir):
    from devo_amd import events
    z = np.load(os.path.join(golden_dir, "events_f32.npz"))
    for tag, (H, W) in (("int", (48, 64)), ("frac", (40, 56))):
        t = lambda k: torch.from_numpy(z[f"{tag}/{k}"]).to(DEV)
        vox = events.to_voxel_grid(t("xs"), t("ys"), t("ts"), t("ps"), H, W, 5)
        ref = torch.from_numpy(z[f"{tag}/vox"])
        assert vox.shape == ref.shape and vox.dtype == torch.float32
        assert_rel(vox, ref, 1e-5, f"{tag} voxel grid")
        seq = torch.stack([ref, ref.flip(0) * 0.5])[None].to(DEV)
        assert_rel(events.std(seq, True), torch.from_numpy(z[f"{tag}/std_seq"]), 1e-5, f"{tag} std sequence")
        assert_rel(events.std(seq, False), torch.from_numpy(z[f"{tag}/std_frame"]), 1e-5, f"{tag} std frame")


def test_voxel_edge_cases():
    from devo_amd import events
    # empty stream -> empty grid; a single event at an integer position lands in one voxel pair (t = 0/0 is NaN: dropped)
    e = lambda *a, dt=torch.float32: torch.tensor(a, dtype=dt, device=DEV)
    z = events.to_voxel_grid(e(), e(), e(dt=torch.float64), e(dt=torch.int8), 8, 9, 5)
    assert z.shape == (5, 8, 9) and float(z.abs().sum()) == 0.0
    g = torch.Generator().manual_seed(5)
    N, H, W = 300000, 120, 160
    xs, ys = torch.rand(N, generator=g) * (W + 4) - 2, torch.rand(N, generator=g) * (H + 4) - 2
    ts = torch.sort(torch.rand(N, generator=g, dtype=torch.float64) * 5e4).values + 1e9
    ps = torch.randint(0, 2, (N,), generator=g).to(torch.int8)
    ref = EV.to_voxel_grid(xs.numpy(), ys.numpy(), ts.numpy(), ps.numpy(), H, W, 5)
    got = events.to_voxel_grid(xs.to(DEV), ys.to(DEV), ts.to(DEV), ps.to(DEV), H, W, 5)
    assert_rel(got, ref, 1e-5, "large stream")
    # std: a segment without events leaves everything unchanged (voxel_utils.py:19)
    seq = torch.stack([ref, torch.zeros_like(ref)])[None].to(DEV)
    assert torch.equal(events.std(seq, sequence=False), seq)
    assert_rel(events.std(seq, sequence=True), EV.std(seq.cpu(), True), 1e-5, "std with an empty frame, sequence-wise")
