"""Patchifier against fixtures produced by the REAL reference modules (tools/gen_golden_patchifier.py:
devo.extractor.BasicEncoder4Evs, devo.selector.Scorer / PatchSelector, devo.enet.Patchifier run on CPU in fp64)."""
import os
import numpy as np
import pytest
import torch

from devo_amd import patchifier as PF
from util import rel_err

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patchifier_f64.npz"))


def _sd(prefix):
    return {k[len(prefix):]: torch.from_numpy(GOLD[k]).double() for k in GOLD.files if k.startswith(prefix)}


def test_encoders_and_scorer_match_the_reference_modules():
    images = torch.from_numpy(GOLD["images"]).double()
    for tag, norm, od in (("fnet", "instance", 16), ("inet", "none", 24)):
        enc = PF.Encoder(5, od, 8, norm).double().eval()
        sd = _sd(tag + "/sd/")
        assert set(sd) == set(enc.state_dict()), "parameter names differ from the reference's state dict"
        enc.load_state_dict(sd)
        with torch.no_grad():
            out = enc(images)
        assert rel_err(out, torch.from_numpy(GOLD[tag + "/out"]).double()) <= 1e-6, tag
    sc = PF.Scorer(5).double().eval()
    sd = _sd("scorer/sd/")
    assert set(sd) == set(sc.state_dict())
    sc.load_state_dict(sd)
    with torch.no_grad():
        assert rel_err(sc(images), torch.from_numpy(GOLD["scorer/out"]).double()) <= 1e-6


def test_parameter_counts_of_the_default_configuration():
    """fnet 184 576 + inet 201 216 + scorer 6 465 (+ Update 3 004 804 = the 3 397 061-parameter bucket)"""
    p = PF.Patchifier()
    cnt = lambda m: sum(q.numel() for q in m.parameters())
    assert (cnt(p.fnet), cnt(p.inet), cnt(p.scorer)) == (184_576, 201_216, 6_465)
    assert set(_sd("pf/sd/")) == set(PF.Patchifier(3, 24, 16, 8, "scorer").state_dict())


def test_pooled_topk_selection_is_the_reference_selection():
    sm = torch.from_numpy(GOLD["topk/scores"]).double()
    for grid in (True, False):
        x, y = PF.select(sm, 8, "topk", grid)
        assert torch.equal(x, torch.from_numpy(GOLD[f"topk/x_grid{int(grid)}"])) and torch.equal(y, torch.from_numpy(GOLD[f"topk/y_grid{int(grid)}"])), grid


def test_stochastic_selections_stay_inside_the_map_and_follow_the_scores():
    g = torch.Generator().manual_seed(3)
    sm = torch.rand(1, 2, 24, 32, generator=g)
    sm[:, :, :, 16:] *= 1e-3                                   # the right half is (almost) never worth a patch
    torch.manual_seed(5)
    for grid in (True, False):
        x, y = PF.select(sm, 16, "multi", grid)
        assert x.shape == (2, 16) and int(x.min()) >= 0 and int(x.max()) < 32 and int(y.min()) >= 0 and int(y.max()) < 24
        if not grid:
            assert float((x < 17).float().mean()) > 0.9
    x, y, s = PF.select_three_x_random(sm, 8)
    assert x.shape == (2, 8) and int(x.min()) >= 1 and int(x.max()) <= 32 and bool((s[:, 1:] >= s[:, :-1]).all())
    assert float((x <= 16).float().mean()) > 0.9


@pytest.mark.gpu
def test_patchifier_forward_matches_the_reference_module():
    """training mode, scorer selection, the reference's own random candidates: every returned tensor"""
    dev = "cuda"
    pf = PF.Patchifier(3, 24, 16, 8, "scorer").to(dev).train()
    pf.load_state_dict({k: v.float() for k, v in _sd("pf/sd/").items()})
    images = torch.from_numpy(GOLD["images"]).to(dev)
    cand = (torch.from_numpy(GOLD["pf/cand_x"]).to(dev), torch.from_numpy(GOLD["pf/cand_y"]).to(dev))
    with torch.no_grad():
        fmap, gmap, imap, patches, index, scores = pf(images, patches_per_image=6, candidates=cand)
    for name, got in dict(fmap=fmap, gmap=gmap, imap=imap, patches=patches, scores=scores).items():
        ref = torch.from_numpy(GOLD["pf/" + name]).float()
        assert got.shape == ref.shape, name
        assert rel_err(got.cpu().float(), ref) <= 2e-4, name
    assert torch.equal(index.cpu(), torch.from_numpy(GOLD["pf/index"]))
    # eval mode: deterministic pooled top-k, colour output, explicit depths
    pf.eval()
    with torch.no_grad():
        disps = torch.rand(1, 2, 12, 16, device=dev) + 0.5
        out = pf(images, patches_per_image=8, disps=disps, return_color=True, scorer_eval_mode="topk")
    fmap, gmap, imap, patches, index, clr = out
    assert patches.shape == (1, 16, 3, 3, 3) and clr.shape == (1, 16, 1)
    x, y = patches[0, :, 0, 1, 1].long(), patches[0, :, 1, 1, 1].long()
    assert torch.allclose(patches[0, :, 2, 1, 1], disps[0, index, y, x])
    assert torch.allclose(gmap[0, :, :, 1, 1], fmap[0, index, :, y, x], atol=1e-6)
    # DEVO's default evaluation selection (config/default.yaml: 'multi' on a 2 x 2 grid): stochastic — shapes, ranges, one patch set per frame
    torch.manual_seed(11)
    with torch.no_grad():
        f2, g2, i2, p2, idx2 = pf(images, patches_per_image=8)
    assert g2.shape == (1, 16, 16, 3, 3) and i2.shape == (1, 16, 24, 1, 1) and p2.shape == (1, 16, 3, 3, 3)
    cx, cy = p2[0, :, 0, 1, 1], p2[0, :, 1, 1, 1]
    assert float(cx.min()) >= 1 and float(cx.max()) <= 16 - 2 and float(cy.min()) >= 1 and float(cy.max()) <= 12 - 2
    assert torch.equal(idx2.cpu(), torch.arange(2).repeat_interleave(8))
    # both lookup levels in the kernel's layout, one pass
    l0, l1 = pf.pyramid(fmap)
    from devo_amd import altcorr
    assert torch.equal(l0, altcorr.channel_blocked(fmap, 8))
    assert rel_err(l1, altcorr.channel_blocked(torch.nn.functional.avg_pool2d(fmap[0], 4, 4)[None], 8)) <= 1e-6


def test_nms_selection_is_the_reference_selection(golden_dir):
    """tests/golden/nms_select.npz (tools/gen_golden_nms.py: the reference's PatchSelector("nms"), its batched_nms a plain greedy loop): pooled
    maxima, boxes, the reference's quadrant categories, per-frame top m — the same coordinates, with and without the grid."""
    z = np.load(os.path.join(golden_dir, "nms_select.npz"))
    for tag in ("a", "b"):
        sm = torch.from_numpy(z[f"{tag}/scores"])
        for grid in (True, False):
            x, y = PF.select(sm, int(z[f"{tag}/m"]), "nms", grid)
            assert torch.equal(x, torch.from_numpy(z[f"{tag}/x_grid{int(grid)}"])) and torch.equal(y, torch.from_numpy(z[f"{tag}/y_grid{int(grid)}"])), (tag, grid)
    with pytest.raises(RuntimeError, match="keeps"):                      # more patches than survivors: a clear error (the reference fails in torch.cat)
        PF.select(torch.from_numpy(z["a/scores"]), 400, "nms", False)


def test_batched_nms_is_the_greedy_suppression():
    """devo_amd.patchifier.batched_nms (whole-vector fixpoint per category) against the one-box-at-a-time greedy loop on random boxes with
    ties, several categories and chains of suppression (a kept, b suppressed by a, c overlapping b only: kept)."""
    g = torch.Generator().manual_seed(2)
    for n, cats, thr in ((60, 1, 0.4), (200, 5, 0.3), (150, 3, 0.0)):
        xy = torch.rand(n, 2, generator=g) * 20
        wh = 2 + 3 * torch.rand(n, 2, generator=g)
        boxes = torch.cat([xy, xy + wh], 1)
        scores = torch.randint(0, 40, (n,), generator=g).float()               # ties
        idxs = torch.randint(0, cats, (n,), generator=g)
        order = sorted(range(n), key=lambda i: (-float(scores[i]), i))
        kept = []
        for i in order:
            def iou(a, b):
                iw = max(0.0, min(float(a[2]), float(b[2])) - max(float(a[0]), float(b[0])))
                ih = max(0.0, min(float(a[3]), float(b[3])) - max(float(a[1]), float(b[1])))
                inter = iw * ih
                return inter / (float((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1])) - inter)
            if all(int(idxs[i]) != int(idxs[j]) or iou(boxes[i], boxes[j]) <= thr for j in kept):
                kept.append(i)
        assert PF.batched_nms(boxes, scores, idxs, thr).tolist() == kept
    chain = torch.tensor([[0.0, 0, 4, 4], [2.0, 0, 6, 4], [4.5, 0, 8.5, 4]])      # a-b overlap, b-c overlap, a-c do not
    assert PF.batched_nms(chain, torch.tensor([3.0, 2.0, 1.0]), torch.zeros(3, dtype=torch.long), 0.2).tolist() == [0, 2]
    assert PF.batched_nms(torch.empty(0, 4), torch.empty(0), torch.empty(0, dtype=torch.long), 0.4).numel() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-6), (torch.float32, 2e-3)])
def test_patchifier_gradients_match_the_reference_module(golden_dir, dt, tol):
    """tests/golden/patchifier_grad_f64.npz (tools/gen_golden_patchifier_grad.py: the reference's Patchifier in training mode on CPU in fp64, a
    loss on fmap, the gathered gmap / imap patches and the winners' scores): the same loss and the same gradients in every parameter of
    both encoders and the scorer — the autograd path through the HIP patch gathers, the channels-last MIOpen encoders and the score lookup."""
    z = np.load(os.path.join(golden_dir, "patchifier_grad_f64.npz"))
    dev = "cuda"
    pf = PF.Patchifier(3, 24, 16, 8, "scorer").to(dev).to(dt).train()
    pf.load_state_dict({k[3:]: torch.from_numpy(z[k]).to(dt) for k in z.files if k.startswith("sd/")})
    images = torch.from_numpy(z["images"]).to(dev).to(dt)
    cand = (torch.from_numpy(z["cand_x"]).to(dev), torch.from_numpy(z["cand_y"]).to(dev))
    fmap, gmap, imap, patches, index, scores = pf(images, patches_per_image=6, candidates=cand)
    for name, got in dict(fmap=fmap, gmap=gmap, imap=imap, patches=patches, scores=scores).items():
        assert rel_err(got.detach().cpu().float(), torch.from_numpy(z["out/" + name]).float().reshape(got.shape)) <= 2e-4, name
    w = {k: torch.from_numpy(z["w/" + k]).to(dev).to(dt) for k in ("fmap", "gmap", "imap", "scores")}
    loss = (fmap * w["fmap"]).sum() * 1e-2 + (gmap.to(dt) * w["gmap"]).sum() + (imap.to(dt) * w["imap"]).sum() + (scores * w["scores"]).sum()
    loss.backward()
    assert abs(float(loss.detach()) - float(z["loss"])) <= tol * abs(float(z["loss"]))
    grads = {k: v.grad for k, v in pf.named_parameters() if v.grad is not None}
    names = [str(s) for s in z["grad_names"]]
    assert sorted(grads) == names
    norms = torch.tensor([float(grads[k].double().norm()) for k in names], dtype=torch.float64)
    assert rel_err(norms, torch.from_numpy(z["grad_norms"])) <= tol, "gradient norms"
    for k in z.files:
        if k.startswith("grad/"):
            assert rel_err(grads[k[5:]].double().cpu(), torch.from_numpy(z[k])) <= tol, k


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_autocast_inference_on_the_cached_low_precision_parameters(dt):
    """devo.py:250 calls patchify under autocast with fp32 parameters, once per frame.  Round 6: that call runs the encoders on a low-precision copy of
    the parameters kept per parameter version instead of letting autocast convert every weight again for every frame — the same arithmetic: every
    output equals autocast's own (DEVO_PATCHIFIER_LOWP=0) to the last bits of the low precision, the copy follows the parameters, the module still
    copies / pickles, and gradients (training) never take this path."""
    import copy
    dev = "cuda"
    torch.manual_seed(3)
    pf = PF.Patchifier().to(dev).eval()
    images = torch.randn(1, 2, 5, 96, 128, device=dev)
    def run(m):
        with torch.no_grad(), torch.autocast("cuda", dtype=dt):
            return m(images, patches_per_image=12, scorer_eval_mode="topk")
    got = run(pf)
    assert "_lowp" in pf.__dict__ or "_lowp_cl" in pf.__dict__
    PF._LOWP = False
    try:
        ref = run(pf)
    finally:
        PF._LOWP = True
    assert got[0].dtype == ref[0].dtype == dt
    tol = 2e-2 if dt == torch.bfloat16 else 4e-3
    assert torch.equal(got[4], ref[4])
    assert rel_err(got[0].float(), ref[0].float()) <= tol and rel_err(got[2].float(), ref[2].float()) <= tol
    same_patches = torch.equal(got[3], ref[3])                                   # (a score tie decided the other way moves a patch: compare its features only then)
    if same_patches:
        assert rel_err(got[1].float(), ref[1].float()) <= tol
    with torch.no_grad():
        pf.fnet.conv2.weight.mul_(0.5)                                           # an optimiser-style step: the copy follows
    got2 = run(pf)
    assert rel_err(got2[0].float(), 0.5 * (got[0].float() - pf.fnet.conv2.bias.to(dt).float()[None, None, :, None, None] / 4) +
                   pf.fnet.conv2.bias.to(dt).float()[None, None, :, None, None] / 4) <= 4 * tol
    c = copy.deepcopy(pf)
    assert "_lowp" not in c.__dict__ and "_lowp_cl" not in c.__dict__ and "_enc_graph" not in c.__dict__ and rel_err(run(c)[0].float(), got2[0].float()) <= tol
    # one frame at a time the encoders replay from a HIP graph from the third call on: the same tensors, and what an earlier call returned stays
    one = images[:, :1].contiguous()
    def run1(m, x):
        with torch.no_grad(), torch.autocast("cuda", dtype=dt):
            return m(x, patches_per_image=12, scorer_eval_mode="topk")
    first = run1(pf, one)
    run1(pf, one)
    third = run1(pf, one)
    assert pf.__dict__["_enc_graph"]["graph"] is not None
    for a, b in zip(first[:4], third[:4]):
        assert torch.equal(a, b)
    keep = third[0].clone()
    other = run1(pf, torch.randn_like(one))
    assert torch.equal(third[0], keep) and not torch.equal(other[0], keep)
    with torch.no_grad():
        pf.inet.conv2.bias.add_(1.0)                                              # new parameter version: a new copy, a new graph later, right values now
    moved = run1(pf, one)
    assert rel_err(moved[2].float() - third[2].float(), torch.full_like(moved[2].float(), 0.25)) <= 4 * tol
    pf.train()
    x = images.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=dt):
        out = pf(x, patches_per_image=12)
    out[0].float().sum().backward()
    assert x.grad is not None and float(x.grad.abs().sum()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.float16, 2e-3)])
def test_fused_instance_norm_of_a_channels_last_activation(dt, tol):
    """devo_instnorm_cl (round 6): relu(instance_norm(x)), instance_norm(x) and the residual block's tail relu(res + relu(instance_norm(x))) on
    channels-last activations, against float64 (extractor.py:27-54: InstanceNorm2d without affine parameters, biased variance, eps 1e-5) — images
    with a large mean (the partial sums are shifted), odd sizes, both channel counts of the encoders; run to run the same bits; with gradients
    enabled the ATen composition runs."""
    dev = "cuda"
    g = torch.Generator().manual_seed(5)
    for N, C, H, W, mean in ((1, 32, 60, 80, 0.0), (3, 64, 31, 45, 0.0), (2, 32, 17, 23, 300.0), (1, 64, 120, 160, -40.0)):
        x = (torch.randn(N, C, H, W, generator=g) * 2.0 + mean).to(dev).to(dt).contiguous(memory_format=torch.channels_last)
        r = torch.randn(N, C, H, W, generator=g).to(dev).to(dt).contiguous(memory_format=torch.channels_last)
        xd = x.double()
        nd = (xd - xd.mean(dim=(2, 3), keepdim=True)) / torch.sqrt(xd.var(dim=(2, 3), unbiased=False, keepdim=True) + 1e-5)
        with torch.no_grad():
            a = PF._in_relu(x)
            b = PF._in_relu(x, relu=False)
            c = PF._in_relu(x, residual=r)
            a2 = PF._in_relu(x)
        assert a.is_contiguous(memory_format=torch.channels_last) and a.dtype == dt
        scale = 1.0 if mean == 0.0 else (1.0 + abs(mean) * (1e-3 if dt == torch.float16 else 0.0))      # (fp16 inputs at 300 carry 0.25 of rounding themselves)
        assert float((b.double() - nd).abs().max()) <= tol * 8 * scale, (N, C, H, W, mean)
        assert float((a.double() - nd.clamp(min=0)).abs().max()) <= tol * 8 * scale
        want_c = (r.double() + nd.clamp(min=0).to(dt).double()).clamp(min=0)
        assert float((c.double() - want_c).abs().max()) <= tol * 16 * scale
        assert torch.equal(a, a2)
        # the convolution's bias added in front, as ATen adds it behind a bias-free convolution: the bits of the norm of (x + bias)
        bias = torch.randn(C, generator=g).to(dev).to(dt)
        xb = x + bias.view(1, -1, 1, 1)
        with torch.no_grad():
            assert torch.equal(PF._in_relu(x, bias=bias), PF._in_relu(xb)) and torch.equal(PF._in_relu(x, residual=r, bias=bias), PF._in_relu(xb, residual=r))
            # ... and the context encoder's epilogues (no norm): bias + ReLU (+ the block's sum and ReLU) in one launch: ATen's bits
            assert torch.equal(PF._bias_act(x, bias=bias), torch.relu(xb))
            assert torch.equal(PF._bias_act(x, relu=False, bias=bias), xb)
            assert torch.equal(PF._bias_act(x, residual=r, bias=bias), torch.relu(r + torch.relu(xb)))
        x2 = x.clone().requires_grad_(True)
        y = PF._in_relu(x2)                                                        # gradients: ATen's kernels
        assert y.requires_grad and float((y.detach().double() - nd.clamp(min=0)).abs().max()) <= tol * 8 * scale
