"""GPU parity of the Update operator (devo_amd/update.py + csrc/update.hip through the C ABI)
against the CPU oracle (oracle/update.py, itself pinned to the reference's devo.enet.Update by tests/golden/update_f64.npz):
the reference-generated golden at dim 32, the real width (dim 384) on a cfg-like graph with random weights, fp16
storage, the torch/autograd path, and the individual fused ops.  Tolerance: 1e-4 of the output scale in fp32 (north_star),
2e-2 for fp16 storage."""
import os
import numpy as np
import pytest
import torch
from oracle import update as U
from devo_amd import synth
from util import assert_rel

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _module(sd, p=3, dim=None, dtype=torch.float32):
    from devo_amd.update import Update
    dim = dim or sd["norm.weight"].numel()
    m = Update(p, dim=dim)
    m.load_state_dict({k: v.float() for k, v in sd.items()})       # the reference's own keys
    return m.to(DEV).to(dtype).eval()


def test_update_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "update_f64.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd/")}
    m = _module(sd)
    for tag in ("irregular", "full"):
        t = lambda k: torch.from_numpy(z[f"{tag}/{k}"])
        with torch.no_grad():
            net, (delta, weight, _) = m(t("net").to(DEV), t("inp").to(DEV), t("corr").to(DEV), None,
                                        t("ii").to(DEV), t("jj").to(DEV), t("kk").to(DEV))
        assert_rel(net, t("net_out"), 1e-4, f"{tag} net")
        assert_rel(delta, t("delta"), 1e-4, f"{tag} delta")
        assert_rel(weight, t("weight"), 1e-4, f"{tag} weight")


def _random_case(n=8, M=12, dim=384, seed=5, keep=0.9):
    from devo_amd.update import Update
    torch.manual_seed(seed)
    m = Update(3, dim=dim)
    sd = {k: v.double() for k, v in m.state_dict().items()}
    ii, jj, kk = synth.full_graph(n, M)
    g = torch.Generator().manual_seed(seed)
    sel = torch.randperm(len(ii), generator=g)[: int(keep * len(ii))]
    ii, jj, kk = ii[sel], jj[sel], kk[sel]
    E = len(ii)
    net = torch.randn(1, E, dim, generator=g)
    inp = torch.randn(1, E, dim, generator=g)
    corr = torch.randn(1, E, 882, generator=g)
    return m, sd, net, inp, corr, ii, jj, kk


def test_update_full_width_fp32_fp16_and_torch_path():
    m, sd, net, inp, corr, ii, jj, kk = _random_case()
    ref = U.update(sd, net.double(), inp.double(), corr.double(), ii, jj, kk)
    args = (net.to(DEV), inp.to(DEV), corr.to(DEV), None, ii.to(DEV), jj.to(DEV), kk.to(DEV))
    m = m.to(DEV).eval()
    with torch.no_grad():
        n1, (d1, w1, _) = m(*args)
        n1b, (d1b, w1b, _) = m(*args)                               # cached graph tables
    for got, want, name in ((n1, ref[0], "net"), (d1, ref[1], "delta"), (w1, ref[2], "weight")):
        assert_rel(got, want, 1e-4, f"fp32 {name}")
    assert torch.equal(n1, n1b) and torch.equal(d1, d1b) and torch.equal(w1, w1b)
    # differentiable torch composition on the GPU (no torch_scatter): same values, gradients flow
    netg = args[0].clone().requires_grad_(True)
    n2, (d2, w2, _) = m(netg, *args[1:])
    assert_rel(n2.detach(), ref[0], 1e-4, "torch path net")
    (d2.sum() + w2.sum()).backward()
    assert torch.isfinite(netg.grad).all() and float(netg.grad.abs().max()) > 0
    # fp16 storage (DEVO runs the update under autocast)
    mh = m.half()
    with torch.no_grad():
        n3, (d3, w3, _) = mh(args[0].half(), args[1].half(), args[2].half(), None, *args[4:])
    assert n3.dtype == torch.float16
    assert_rel(n3.float(), ref[0], 2e-2, "fp16 net")
    assert_rel(w3.float(), ref[2], 2e-2, "fp16 weight")


def test_update_under_autocast_with_gradients_enabled():
    """devo.py:311 runs the update operator under torch.autocast; with fp32 parameters that require grad (any caller that does not
    wrap inference in no_grad, or mixed-precision fine-tuning) the module takes its autograd path: LayerNorm outputs stay fp32 while
    the Linear outputs are fp16.  The custom Functions must cope with the mix — values near the fp32 result, finite gradients in the
    parameters' dtype."""
    m, sd, net, inp, corr, ii, jj, kk = _random_case()
    ref = U.update(sd, net.double(), inp.double(), corr.double(), ii, jj, kk)
    m = m.to(DEV).train()
    args = (net.to(DEV).requires_grad_(True), inp.to(DEV), corr.to(DEV), None, ii.to(DEV), jj.to(DEV), kk.to(DEV))
    with torch.autocast("cuda", dtype=torch.float16):
        n, (d, w, _) = m(*args)
        loss = d.float().sum() + w.float().sum() + n.float().pow(2).mean()
    loss.backward()
    assert_rel(n.float(), ref[0], 3e-2, "autocast net")
    assert_rel(w.float(), ref[2], 3e-2, "autocast weight")
    assert args[0].grad.dtype == torch.float32 and bool(torch.isfinite(args[0].grad).all()) and float(args[0].grad.abs().max()) > 0
    for name, p in m.named_parameters():
        if p.grad is not None:
            assert p.grad.dtype == p.dtype and bool(torch.isfinite(p.grad).all()), name
    assert sum(p.grad is not None for p in m.parameters()) > 10


def test_update_graph_tables_follow_a_changing_graph():
    """DEVO rebuilds ii / jj / kk with torch.cat every frame (devo.py:392-399): fresh tensors of often the same size, whose
    storage the caching allocator may hand back at the SAME address.  The cached neighbour / group tables must follow the
    graph, and the frame-pair grouping must not scale with the absolute frame index."""
    from devo_amd.update import Update
    m, sd, net, inp, corr, ii, jj, kk = _random_case(n=6, M=10, dim=64, seed=9)
    m = m.to(DEV).eval()
    E = len(ii)
    g = torch.Generator().manual_seed(3)
    outs, refs = [], []
    for step in range(4):
        perm = torch.randperm(E, generator=g)                       # another graph over the same number of edges
        a, b, c = ii[perm] + 4000 * step, jj[perm] + 4000 * step, kk[perm]      # (long sequences: large absolute frame ids)
        refs.append(U.update(sd, net.double(), inp.double(), corr.double(), a, b, c))
        ta, tb, tc = a.to(DEV), b.to(DEV), c.to(DEV)                # fresh device tensors, the previous ones are freed below
        with torch.no_grad():
            n1, (d1, w1, _) = m(net.to(DEV), inp.to(DEV), corr.to(DEV), None, ta, tb, tc)
        outs.append((n1.cpu(), d1.cpu(), w1.cpu()))
        del ta, tb, tc
    for (n1, d1, w1), ref in zip(outs, refs):
        assert_rel(n1, ref[0], 1e-4, "net"); assert_rel(d1, ref[1], 1e-4, "delta"); assert_rel(w1, ref[2], 1e-4, "weight")
    # in-place edits of the same tensors are seen through the version counter
    ta, tb, tc = ii.to(DEV), jj.to(DEV), kk.to(DEV)
    with torch.no_grad():
        m(net.to(DEV), inp.to(DEV), corr.to(DEV), None, ta, tb, tc)
        perm = torch.randperm(E, generator=g).to(DEV)
        ta.copy_(ta[perm]); tb.copy_(tb[perm]); tc.copy_(tc[perm])
        n2, (d2, w2, _) = m(net.to(DEV), inp.to(DEV), corr.to(DEV), None, ta, tb, tc)
    ref = U.update(sd, net.double(), inp.double(), corr.double(), ta.cpu(), tb.cpu(), tc.cpu())
    assert_rel(n2, ref[0], 1e-4, "net after an in-place edit")


def test_update_single_ops():
    """layer norm (+ fused adds, ReLU), masked gather, soft aggregation, gated residual, heads — each against torch"""
    import devo_amd._lib as L
    from devo_amd.update import _Groups
    g = torch.Generator().manual_seed(3)
    E, dim = 777, 384
    x, a, b = (torch.randn(E, dim, generator=g).to(DEV) for _ in range(3))
    gam, bet = torch.randn(dim, generator=g).to(DEV), torch.randn(dim, generator=g).to(DEV)
    out = torch.empty_like(x)
    lib = L.lib()
    N0 = None
    L.check(lib.devo_upd_layernorm(L.ptr(x), L.ptr(a), L.ptr(b), N0, N0, N0, 0, N0, L.ptr(gam), L.ptr(bet), L.ptr(out), E, dim, 1e-3, 1, 0,
                                   L.stream()), "ln")
    ref = torch.relu(torch.nn.functional.layer_norm((x + a + b).double(), (dim,), gam.double(), bet.double(), 1e-3))
    assert_rel(out, ref, 1e-5, "layernorm")
    # all fused input terms at once: + hy[group_of] + sigmoid(gate) * res, gate with a wider row stride
    hy = torch.randn(50, dim, generator=g).to(DEV)
    grp = torch.randint(0, 50, (E,), generator=g).to(torch.int32).to(DEV)
    wide = torch.randn(E, 2 * dim, generator=g).to(DEV)
    L.check(lib.devo_upd_layernorm(L.ptr(x), L.ptr(a), N0, L.ptr(hy), L.ptr(grp), L.ptr(wide), 2 * dim, L.ptr(b), L.ptr(gam), L.ptr(bet),
                                   L.ptr(out), E, dim, 1e-3, 0, 0, L.stream()), "ln fused")
    pre = x.double() + a.double() + hy.double()[grp.long()] + torch.sigmoid(wide[:, :dim].double()) * b.double()
    assert_rel(out, torch.nn.functional.layer_norm(pre, (dim,), gam.double(), bet.double(), 1e-3), 1e-5, "fused layernorm")
    idx = torch.randint(-1, E, (E,), generator=g).to(DEV)
    L.check(lib.devo_upd_masked_gather(L.ptr(x), L.ptr(idx), L.ptr(out), E, dim, 0, L.stream()), "gather")
    assert torch.equal(out, x[idx] * (idx >= 0).float()[:, None])
    key = torch.randint(0, 40, (E,), generator=g) * 977                # sparse key range
    G = _Groups(key.to(DEV))
    y = torch.empty(G.n_seg, dim, device=DEV)
    L.check(lib.devo_upd_softagg(L.ptr(x), L.ptr(a), dim, L.ptr(G.perm), L.ptr(G.seg_start), L.ptr(G.n_seg_dev), L.ptr(y), L.ptr(G.group_of),
                                 E, dim, 0, L.stream()), "softagg")
    _, inv = torch.unique(key, return_inverse=True)
    yref = U.segment_softmax_sum(x.cpu().double()[None], a.cpu().double()[None], inv)[0]
    assert G.n_seg == yref.shape[0] and torch.equal(G.group_of.cpu().long(), inv)
    assert_rel(y, yref, 1e-5, "softagg")
    net = b.clone()
    L.check(lib.devo_upd_expand_add(L.ptr(net), L.ptr(y), L.ptr(G.group_of), E, dim, 0, L.stream()), "expand")
    assert_rel(net, b.cpu().double() + yref[inv], 1e-5, "expand_add")
    L.check(lib.devo_upd_gated_residual(L.ptr(x), L.ptr(a), dim, L.ptr(b), L.ptr(out), E, dim, 0, L.stream()), "gated")
    assert_rel(out, x.double() + torch.sigmoid(a.double()) * b.double(), 1e-5, "gated residual")
    Wd, Ww = torch.randn(2, dim, generator=g).to(DEV) / 20, torch.randn(2, dim, generator=g).to(DEV) / 20
    bd, bw = torch.randn(2, generator=g).to(DEV), torch.randn(2, generator=g).to(DEV)
    delta, weight = torch.empty(E, 2, device=DEV), torch.empty(E, 2, device=DEV)
    L.check(lib.devo_upd_heads(L.ptr(x), N0, 0, N0, N0, L.ptr(Wd), L.ptr(bd), L.ptr(Ww), L.ptr(bw), L.ptr(delta), L.ptr(weight), E, dim, 0,
                               L.stream()), "heads")
    r = torch.relu(x.double())
    assert_rel(delta, r @ Wd.double().t() + bd.double(), 1e-5, "delta head")
    assert_rel(weight, torch.sigmoid(r @ Ww.double().t() + bw.double()), 1e-5, "weight head")
    net_out = torch.empty_like(x)
    L.check(lib.devo_upd_heads(L.ptr(x), L.ptr(wide), 2 * dim, L.ptr(b), L.ptr(net_out), L.ptr(Wd), L.ptr(bd), L.ptr(Ww), L.ptr(bw),
                               L.ptr(delta), L.ptr(weight), E, dim, 0, L.stream()), "gated heads")
    nref = x.double() + torch.sigmoid(wide[:, :dim].double()) * b.double()
    assert_rel(net_out, nref, 1e-5, "gated residual inside the heads kernel")
    assert_rel(delta, torch.relu(nref) @ Wd.double().t() + bd.double(), 1e-5, "delta head after the gated residual")


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float16, 2e-3)])
def test_softagg_one_wave_per_group(dtype, tol):
    """k_softagg_w (round 6): small groups — a patch's edges — by one wave each, the rows requested together, exact batch maxima, batches of a
    longer group merged online.  Groups of 1 ... 40 rows (1, 15, 16, 17, 33 among them: below / at / above one and two batches of fp16 and fp32),
    against the float64 segment softmax (blocks.py:42-43) and against the four-wave form of the same launch (hint 0)."""
    from devo_amd.update import _Groups
    from devo_amd import _lib as L
    g = torch.Generator().manual_seed(11)
    sizes = [1, 15, 16, 17, 33, 40, 8, 9, 2] + [int(v) for v in torch.randint(1, 30, (150,), generator=g)]
    key = torch.cat([torch.full((n,), 7 * i + 3, dtype=torch.int64) for i, n in enumerate(sizes)])
    key = key[torch.randperm(key.numel(), generator=g)]
    E, dim = key.numel(), 384
    x = (torch.randn(E, 2 * dim, generator=g) * 2.0).to(DEV).to(dtype)              # f | g
    G = _Groups(key.to(DEV))
    lib, code = L.lib(), L.dtype_code(x)
    outs = []
    for hint in (E // len(sizes), 0):
        y = torch.zeros(G.n_seg, dim, device=DEV, dtype=dtype)
        grp = torch.full((E,), -1, dtype=torch.int32, device=DEV)
        L.check(lib.devo_upd_softagg_hint(L.ptr(x), L.ptr(x[:, dim:]), 2 * dim, L.ptr(G.perm), L.ptr(G.seg_start), L.ptr(G.n_seg_dev), L.ptr(y), L.ptr(grp),
                                          E, dim, code, hint, L.stream()), "softagg")
        outs.append((y, grp))
    _, inv = torch.unique(key, return_inverse=True)
    yref = U.segment_softmax_sum(x[:, :dim].cpu().double()[None], x[:, dim:].cpu().double()[None], inv)[0]
    assert G.n_seg == len(sizes) == yref.shape[0]
    for y, grp in outs:
        assert torch.equal(grp.cpu().long(), inv)
        assert_rel(y.float(), yref, tol, "softagg")
    assert_rel(outs[0][0].float(), outs[1][0].double(), tol, "one wave per group against four")


def test_gradient_clip_backward_semantics():
    """blocks.py:72-81: the delta / weight heads pass gradients through GradClip — NaN -> 0, clamped to +-0.01"""
    from devo_amd.update import GradientClip
    x = torch.tensor([1.0, 2.0, 3.0, 4.0], device=DEV, requires_grad=True)
    y = GradientClip()(x)
    assert torch.equal(y, x)
    y.backward(torch.tensor([0.5, -0.5, float("nan"), 0.003], device=DEV))
    assert torch.equal(x.grad.cpu(), torch.tensor([0.01, -0.01, 0.0, 0.003]))


def test_softagg_training_path_on_the_hip_kernels_matches_the_torch_composition():
    """SoftAgg in the differentiable path (blocks.py:31-48): f | g in one GEMM, devo_upd_softagg / devo_upd_softagg_backward, h, expand —
    against the torch composition (scatter_reduce amax / exp / scatter_add / gather) of the same module: output and the gradients of
    the input and of all six parameter tensors; irregular groups (sizes 1 .. 40), an edge count that is no multiple of anything."""
    from devo_amd.update import SoftAgg, _Groups
    torch.manual_seed(3)
    dim, E = 64, 4999
    key = torch.randint(0, 700, (E,), device=DEV)
    agg = SoftAgg(dim).to(DEV)
    x0 = torch.randn(1, E, dim, device=DEV)
    wgt = torch.randn(1, E, dim, device=DEV)
    G = _Groups(key.long().contiguous())

    def run(groups):
        for q in agg.parameters():
            q.grad = None
        x = x0.clone().requires_grad_(True)
        y = agg(x, key, groups)
        (y * wgt).sum().backward()
        return [y.detach(), x.grad] + [q.grad.clone() for q in agg.parameters()]

    a, b = run(G), run(None)
    names = ["output", "d/dx"] + ["d/d" + n for n, _ in agg.named_parameters()]
    scale = float(a[names.index("d/df.bias")].abs().max())
    for u, v, n in zip(a, b, names):
        if n == "d/dg.bias":                                      # softmax is shift-invariant inside a group: this gradient is exactly 0
            assert float(u.abs().max()) <= 1e-4 * scale and float(v.abs().max()) <= 1e-4 * scale
        else:
            assert_rel(u, v, 2e-4, n)
    # the whole operator: forward_torch on the HIP tables equals forward_torch on torch.unique maps (fp64 takes the latter)
    from devo_amd.update import Update
    n, M = 5, 6
    ii, jj, kk = [t.to(DEV) for t in synth.full_graph(n, M)]
    Eg = ii.numel()
    up = Update(3, dim=32).to(DEV)
    net = torch.randn(1, Eg, 32, device=DEV); inp = torch.randn(1, Eg, 32, device=DEV) * 0.1; corr = torch.randn(1, Eg, 882, device=DEV)
    up64 = Update(3, dim=32).to(DEV).double(); up64.load_state_dict({k: v.double() for k, v in up.state_dict().items()})
    wn = torch.randn(1, Eg, 32, device=DEV)

    def run(mod, dt):
        for q in mod.parameters():
            q.grad = None
        a, c = net.detach().to(dt).clone().requires_grad_(True), corr.detach().to(dt).clone().requires_grad_(True)
        o = mod.forward_torch(a, inp.to(dt), c, ii, jj, kk)
        ((o[0] * wn.to(dt)).sum() + o[1][0].sum() + o[1][1].sum()).backward()
        return o, a.grad, c.grad, {k: q.grad for k, q in mod.named_parameters()}

    (o32, ga32, gc32, gp32), (o64, ga64, gc64, gp64) = run(up, torch.float32), run(up64, torch.float64)
    assert_rel(o32[0].detach(), o64[0].detach().float(), 1e-4, "net")
    assert_rel(o32[1][0].detach(), o64[1][0].detach().float(), 1e-3, "delta"); assert_rel(o32[1][1].detach(), o64[1][1].detach().float(), 1e-4, "weight")
    # gradients through the HIP autograd functions (gated residual, masked gather, SoftAgg, split-K Linear) vs the fp64 torch path
    assert_rel(ga32, ga64.float(), 2e-3, "d/d net"); assert_rel(gc32, gc64.float(), 2e-3, "d/d corr")
    for k in ("gru.1.gate.0.weight", "gru.3.res.2.weight", "c1.0.weight", "c2.2.bias", "agg_kk.f.weight", "agg_ij.h.weight", "norm.weight", "corr.0.weight"):
        assert_rel(gp32[k], gp64[k].float(), 3e-3, "d/d " + k)


@pytest.mark.parametrize("rows,n_out,k_in,relu", [(18000, 384, 384, False), (4099, 768, 384, True), (1030, 192, 32, False), (5000, 384, 768, False),
                                                   (3001, 384, 882, True), (2000, 96, 50, False), (3001, 882, 384, False), (2500, 100, 384, False)])
def test_split_precision_linear_matches_float64(rows, n_out, k_in, relu):
    """csrc/linear.hip: fp32 in, fp32 out, fp16 hi + lo operands on the matrix cores — as close to the float64 product as the library's
    fp32 GEMM is (the tolerance is the fp32 GEMM's own distance to float64, doubled), in the forward form and in the dX form"""
    from devo_amd import update as UA
    g = torch.Generator(device="cpu").manual_seed(rows + n_out)
    x = torch.randn(rows, k_in, generator=g) * torch.rand(rows, 1, generator=g) * 4
    x[1::7] *= 1e-7                                                     # gradient-sized rows: below fp16's range before scaling
    x[2::11] *= 3e5
    x[3::13, k_in // 2:] *= 4096.0                                      # rows whose scale is raised in the middle of K ...
    x[5::17, :k_in // 2] = 0                                            # ... or that start with zeros
    x[6::19] = 0
    x = x.to(DEV)
    w = torch.randn(n_out, k_in, generator=g) / k_in ** 0.5
    w[1::5] *= 1e-6
    w[2::9] *= 1e3
    w = w.to(DEV)
    b = torch.randn(n_out, generator=g).to(DEV)
    assert UA._split_ok(x, n_out, k_in)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    lib = torch.nn.functional.linear(x, w, b)
    if relu:
        ref, lib = ref.relu(), lib.relu()
    y = UA._linear_split(x, w, b, relu=relu)
    scale = (x.double().abs() @ w.double().abs().t() + b.double().abs())     # every output against its own terms' magnitudes
    e_lib = ((lib.double() - ref).abs() / scale).max().item()
    e_own = ((y.double() - ref).abs() / scale).max().item()
    assert e_own <= max(2 * e_lib, 2e-7), (e_own, e_lib)
    # dX form: g [rows, n_out] @ w [n_out, k_in], when the transposed sizes fit the kernel
    if k_in % 192 == 0 and n_out % 32 == 0:
        gy = torch.randn(rows, n_out, generator=g).to(DEV)
        ref = gy.double() @ w.double()
        gy[1::3] *= 1e-8
        ref = gy.double() @ w.double()
        scale_t = gy.double().abs() @ w.double().abs() + 1e-300
        e_lib = (((gy @ w).double() - ref).abs() / scale_t).max().item()
        e_own = ((UA._linear_split(gy, w, None, transposed=True).double() - ref).abs() / scale_t).max().item()
        assert e_own <= max(2 * e_lib, 2e-7), (e_own, e_lib)
    # residual added in place, the ReLU from a column on (a GatedResidual's gate | res[0] pair)
    acc = torch.randn(rows, n_out, generator=g).to(DEV)
    ref = acc.double() + torch.nn.functional.linear(x.double(), w.double(), b.double())
    got = UA._linear_split(x, w, b, residual=acc, out=acc)
    assert got.data_ptr() == acc.data_ptr() and ((got.double() - ref).abs() / (scale + ref.abs())).max().item() < 1e-6
    half = n_out // 2 // 4 * 4
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref[:, half:].relu_()
    assert ((UA._linear_split(x, w, b, relu_from=half).double() - ref).abs() / scale).max().item() < 1e-6
    gate = torch.randn(rows, n_out, generator=g).to(DEV)                # a ReLU's output: the result passes where it is positive
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double()) * (gate > 0)
    assert ((UA._linear_split(x, w, b, gate=gate).double() - ref).abs() / scale).max().item() < 1e-6
    poison = torch.full((rows, n_out + 3), 7.0, device=DEV)            # nothing is written behind column N (rows that are not 16-byte aligned)
    UA._linear_split(x, w, b, out=poison[:, :n_out])
    assert torch.equal(poison[:, n_out:], torch.full((rows, 3), 7.0, device=DEV))
    assert ((poison[:, :n_out].double() - torch.nn.functional.linear(x.double(), w.double(), b.double())).abs() / scale).max().item() < 1e-6
    # strided rows (a column slice of a wider tensor) and the per-version cache
    wide = torch.randn(rows, k_in + 67, generator=g).to(DEV)
    xs = wide[:, 33:33 + k_in]                                          # (rows at odd multiples of 4 bytes)
    if UA._split_ok(xs, n_out, k_in):
        ref = torch.nn.functional.linear(xs.double(), w.double(), b.double())
        assert ((UA._linear_split(xs, w, b).double() - ref).abs() / (xs.double().abs() @ w.double().abs().t() + b.double().abs())).max().item() < 1e-6
    n_cached = len(UA._wsplit_cache)
    UA._linear_split(x, w, b)
    assert len(UA._wsplit_cache) == n_cached
    w.mul_(2.0)                                                         # an optimiser step: new version, new image
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    assert ((UA._linear_split(x, w, b).double() - ref).abs() / (x.double().abs() @ w.double().abs().t() + b.double().abs())).max().item() < 1e-6


def test_linear_layer_gradients_with_and_without_the_split_gemm():
    from devo_amd import update as UA
    torch.manual_seed(3)
    lin = UA.Linear(384, 384).to(DEV)
    x = torch.randn(1, 6000, 384, device=DEV, requires_grad=True)
    outs = []
    for flag in (True, False):
        UA.SPLIT_GEMM = flag
        try:
            lin.zero_grad(); x.grad = None
            y = torch.relu_(lin(x))
            (y.square().mean() + y.sum() * 1e-3).backward()
            outs.append((y.detach().clone(), x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone()))
        finally:
            UA.SPLIT_GEMM = True
    for a, b in zip(*outs):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-6 * max(1.0, b.abs().max().item())), (a - b).abs().max()


@pytest.mark.parametrize("rows,relu,adds", [(18000, False, 2), (4099, True, 0), (37, False, 1)])
def test_layernorm_backward_kernel_matches_autograd(rows, relu, adds):
    """devo_upd_layernorm_backward against torch.autograd through the fp64 composition (enet.py:44,52-56,62 differentiate nn.LayerNorm)"""
    from devo_amd import update as UA
    torch.manual_seed(rows)
    mod = UA.LayerNorm(384, eps=1e-3).to(DEV)
    with torch.no_grad():
        mod.weight.copy_(torch.randn(384) * 0.5 + 1.0); mod.bias.copy_(torch.randn(384) * 0.3)
    xs = [torch.randn(1, rows, 384, device=DEV, requires_grad=True) for _ in range(1 + adds)]
    xs[0].data.mul_(torch.rand(1, rows, 1, device=DEV) * 5 + 0.01)
    gout = torch.randn(1, rows, 384, device=DEV)
    y = UA._ln_train(UA._PlainLN(mod), xs[0], *(xs[1:] + [None] * (2 - adds)), relu=relu)
    y.backward(gout)
    got = [y.detach()] + [t.grad.clone() for t in xs] + [mod.weight.grad.clone(), mod.bias.grad.clone()]
    xd = [t.detach().double().requires_grad_(True) for t in xs]
    wd, bd = mod.weight.detach().double().requires_grad_(True), mod.bias.detach().double().requires_grad_(True)
    yd = torch.nn.functional.layer_norm(sum(xd), (384,), wd, bd, 1e-3)
    if relu:
        yd = yd.relu()
    yd.backward(gout.double())
    ref = [yd.detach()] + [t.grad for t in xd] + [wd.grad, bd.grad]
    for name, a, b in zip(["y"] + ["dx"] * len(xs) + ["dgamma", "dbeta"], got, ref):
        err = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
        assert err < (2e-5 if name in ("dgamma", "dbeta") else 5e-6), (name, err)


def test_update_training_path_with_and_without_the_hip_layernorm():
    from devo_amd import update as UA
    torch.manual_seed(11)
    n, M = 6, 30
    ii, jj, kk = [t.to(DEV) for t in synth.full_graph(n, M)]
    E = ii.numel()
    up = UA.Update(3).to(DEV).train()
    net = torch.randn(1, E, 384, device=DEV, requires_grad=True)
    inp = torch.randn(1, E, 384, device=DEV) * 0.1
    corr = torch.randn(1, E, 882, device=DEV, requires_grad=True)
    outs = []
    for flag in (True, False):
        UA.HIP_LAYERNORM = flag
        try:
            up.zero_grad(); net.grad = None; corr.grad = None
            out, (d, w, _) = up(net, inp, corr, None, ii, jj, kk)
            (out.square().mean() + d.square().mean() + w.mean()).backward()
            outs.append([out.detach().clone(), d.detach().clone(), net.grad.clone(), corr.grad.clone()] + [p.grad.clone() for p in up.parameters()])
        finally:
            UA.HIP_LAYERNORM = True
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= 2e-4 * max(b.abs().max().item(), 1e-6) + 1e-7


@pytest.mark.parametrize("rows,n_out,n_in", [(18000, 384, 384), (4099, 768, 384), (2050, 128, 256), (6001, 384, 768), (5003, 384, 882), (2500, 200, 130)])
def test_split_precision_weight_gradient_matches_float64(rows, n_out, n_in):
    """csrc/linear_dw.hip: dW = dY^T X and db = column sums of dY, fp32 in and out on the fp16 matrix cores, against float64 — as close as
    the library's fp32 product (every entry against the magnitudes of its own terms), with gradient-sized columns, columns whose scale
    is raised late, zero columns and a row count that is not a multiple of the step"""
    from devo_amd import update as UA
    g = torch.Generator(device="cpu").manual_seed(rows + n_out)
    dy = torch.randn(rows, n_out, generator=g) * torch.rand(1, n_out, generator=g)
    dy[:, 1::7] *= 1e-9
    dy[:, 2::11] *= 1e4
    dy[rows // 2:, 3::13] *= 4096.0
    dy[:rows // 2, 5::17] = 0
    dy[:, 6::19] = 0
    x = torch.randn(rows, n_in, generator=g) * (torch.rand(1, n_in, generator=g) * 3)
    x[:, 1::5] *= 1e-5
    x[rows // 3:, 2::9] *= 1e3
    dy, x = dy.to(DEV), x.to(DEV)
    assert UA._dw_ok(dy, x)
    dW, db = UA._dw_split(dy, x, True)
    ref = dy.double().t() @ x.double()
    scale = dy.double().abs().t() @ x.double().abs() + 1e-300
    e_lib = (((dy.t() @ x).double() - ref).abs() / scale).max().item()
    e_own = ((dW.double() - ref).abs() / scale).max().item()
    assert e_own <= max(2 * e_lib, 3e-7), (e_own, e_lib)
    rb = dy.double().sum(0)
    assert ((db.double() - rb).abs() / (dy.double().abs().sum(0) + 1e-300)).max().item() < 1e-6
    dW2, _ = UA._dw_split(dy, x, False)
    assert torch.equal(dW, dW2)                                         # no atomics: the same bits every time


@pytest.mark.parametrize("rows,n_out,k_in", [(21600, 384, 384), (4099, 768, 384), (3001, 384, 882), (1030, 96, 70)])
def test_fp16_linear_kernel_matches_the_fp32_product_of_the_same_values(rows, n_out, k_in):
    """csrc/linear.hip k_linear_f16: fp16 rows and weights, fp32 accumulation — against the fp32 product of the SAME fp16 values, to the
    rounding of the fp16 result; bias, ReLU from a column on, residual in place, strided input rows"""
    from devo_amd import update as UA
    g = torch.Generator(device="cpu").manual_seed(rows + k_in)
    x = (torch.randn(rows, k_in, generator=g) * 0.7).half().to(DEV)
    w = (torch.randn(n_out, k_in, generator=g) / k_in ** 0.5).half().to(DEV)
    b = torch.randn(n_out, generator=g).half().to(DEV)
    assert UA._f16_ok(x, n_out, k_in)
    ref = torch.nn.functional.linear(x.float(), w.float(), b.float())
    y = UA._linear_f16(x, w, b)
    assert (y.float() - ref).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-3
    half = n_out // 2 // 4 * 4
    r2 = ref.clone(); r2[:, half:].relu_()
    assert (UA._linear_f16(x, w, b, relu_from=half).float() - r2).abs().max().item() <= 2e-3 * ref.abs().max().item() + 1e-3
    acc = torch.randn(rows, n_out, generator=g).half().to(DEV)
    r3 = acc.float() + ref
    got = UA._linear_f16(x, w, b, residual=acc, out=acc)
    assert got.data_ptr() == acc.data_ptr() and (got.float() - r3).abs().max().item() <= 2e-3 * r3.abs().max().item() + 1e-3
    wide = (torch.randn(rows, k_in + 66, generator=g) * 0.7).half().to(DEV)
    xs = wide[:, 34:34 + k_in]
    assert UA._f16_ok(xs, n_out, k_in)
    r4 = torch.nn.functional.linear(xs.float(), w.float(), b.float())
    assert (UA._linear_f16(xs, w, b).float() - r4).abs().max().item() <= 2e-3 * r4.abs().max().item() + 1e-3


def test_layernorm_parameter_gradients_are_reproducible_in_deterministic_mode():
    """torch.use_deterministic_algorithms(True): gamma / beta gradients of the HIP LayerNorm come from a fixed-order two-stage reduction
    (devo_upd_layernorm_backward with its scratch) — bit-identical from run to run, and equal to the atomic form within rounding."""
    from devo_amd import update as UA
    torch.manual_seed(3)
    mod = UA.LayerNorm(384, eps=1e-3).to(DEV)
    x = torch.randn(1, 18000, 384, device=DEV) * 3
    g = torch.randn(1, 18000, 384, device=DEV)

    def grads():
        mod.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        UA._ln_train(UA._PlainLN(mod), xi, None, None, relu=True).backward(g)
        return mod.weight.grad.clone(), mod.bias.grad.clone()
    atomic = grads()
    was = torch.are_deterministic_algorithms_enabled()
    try:
        torch.use_deterministic_algorithms(True)
        runs = [grads() for _ in range(4)]
    finally:
        torch.use_deterministic_algorithms(was)
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1])
    for a, b in zip(atomic, runs[0]):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()


def test_weight_images_follow_versions_and_invalidate_weights_covers_data_edits():
    """The Linear layers' operand images (csrc/linear.hip) are cached per VERSION of the weight: copy_ is seen, an edit through `.data` is
    not — Update.invalidate_weights() (also called by train() / eval() / load_state_dict) makes it seen."""
    from devo_amd import update as UA
    torch.manual_seed(0)
    lin = torch.nn.Linear(384, 384).to(DEV)
    x = torch.randn(4096, 384, device=DEV)
    y0 = UA._linear_split(x, lin.weight, lin.bias)
    with torch.no_grad():
        lin.weight.copy_(lin.weight * 2.0)                            # version bump: seen
    y1 = UA._linear_split(x, lin.weight, lin.bias)
    assert (y1 - lin.bias - 2.0 * (y0 - lin.bias)).abs().max().item() < 1e-3
    lin.weight.data.mul_(0.5)                                         # .data edit: its own version counter
    y_stale = UA._linear_split(x, lin.weight, lin.bias)
    assert torch.equal(y_stale, y1)                                   # (the documented blind spot)
    UA.invalidate_weight_images()
    y2 = UA._linear_split(x, lin.weight, lin.bias)
    assert (y2 - y0).abs().max().item() < 1e-3
    upd = UA.Update(3).to(DEV)
    upd.eval()                                                        # train(False) clears the caches
    assert len(UA._wsplit_cache) == 0


@pytest.mark.parametrize("K1,rows,gather,residual", [(384, 21600, False, False), (384, 18000, True, True), (882, 21600, False, False), (882, 4099, False, True),
                                                      (384, 1031, True, False)])
def test_fused_linear_relu_linear_matches_the_two_layer_composition(K1, rows, gather, residual):
    """csrc/mlp2.hip (Linear - ReLU - Linear in one launch, the 384-wide intermediate in LDS; fp16 storage): against the fp64 composition
    on the same fp16 operands with the intermediate rounded to fp16 (what two launches of the fp16 Linear kernel compute), every output
    against the magnitude of ITS terms; gathered rows (negative index = zero row), the residual sum, row counts that are no multiple of
    the 64-row tile, the corr MLP's 882 inputs (rows 4-byte aligned only, a masked last K step)."""
    from devo_amd import update as UA
    torch.manual_seed(K1 + rows)
    l1, l2 = torch.nn.Linear(K1, 384).to(DEV).half(), torch.nn.Linear(384, 384).to(DEV).half()
    src_rows = rows if not gather else rows + 37
    x = (torch.randn(src_rows, K1, device=DEV) * 0.7).half()
    idx = None
    if gather:
        idx = torch.randint(0, src_rows, (rows,), device=DEV)
        idx[torch.rand(rows, device=DEV) < 0.1] = -1
    res = (torch.randn(rows, 384, device=DEV)).half() if residual else None
    assert UA._mlp2_ok(x, l1, l2)
    got = UA._mlp2_f16(x, l1, l2, residual=res, gather=idx)
    xs = x.double() if idx is None else torch.where((idx >= 0)[:, None], x.double()[idx.clamp(min=0)], torch.zeros(1, dtype=torch.float64, device=DEV))
    h = torch.relu(xs @ l1.weight.double().t() + l1.bias.double()).half().double()
    ref = h @ l2.weight.double().t() + l2.bias.double()
    mag = h.abs() @ l2.weight.double().abs().t() + l2.bias.double().abs()
    if res is not None:
        ref, mag = ref + res.double(), mag + res.double().abs()
    err = ((got.double() - ref).abs() / mag.clamp(min=1e-6)).max().item()
    assert torch.isfinite(got).all() and err < 2e-3, err          # fp16 rounding of the result (2^-11) + of the intermediate's neighbours


def test_update_fp16_operator_uses_the_fused_chains_and_keeps_its_golden():
    """The fp16 inference operator with and without csrc/mlp2.hip (DEVO_UPD_MLP2): the same outputs within fp16 rounding."""
    from devo_amd import update as UA
    torch.manual_seed(5)
    E, n, M = 21600, 15, 96
    from devo_amd import synth
    ii, jj, kk = [t.to(DEV) for t in synth.full_graph(n, M)]
    upd = UA.Update(3).to(DEV).half().eval()
    net = torch.randn(1, E, 384, device=DEV).half() * 0.1
    inp = torch.randn(1, E, 384, device=DEV).half() * 0.1
    corr = torch.randn(1, E, 882, device=DEV).half()
    outs = []
    for on in (True, False):
        UA.MLP2_F16 = on
        try:
            with torch.no_grad():
                o, (d, w, _) = upd(net, inp, corr, None, ii, jj, kk)
        finally:
            UA.MLP2_F16 = True
        outs.append((o.float(), d.float(), w.float()))
    for a, b in zip(*outs):
        assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 2e-2 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("rows,n_out,relu_from,residual", [(21600, 384, None, False), (18000, 384, 0, True), (21600, 768, 384, False), (1031, 384, None, True),
                                                           (97, 768, 0, False)])
def test_row_resident_linear_matches_the_fp64_product(rows, n_out, relu_from, residual):
    """csrc/gemm_rs.hip (a workgroup's rows in LDS once, the weights from the L2 into the registers of the wave that multiplies them; fp16
    storage, fp32 accumulation) against the fp64 product of the same fp16 operands — every output against the magnitude of its terms —
    and against csrc/linear.hip's kernel (fp16 rounding of the same sums); ReLU from a column on, the in-place residual, row counts
    that are no multiple of the 96-row tile, both column passes of a 768-wide layer."""
    from devo_amd import update as UA
    torch.manual_seed(rows + n_out)
    lin = torch.nn.Linear(384, n_out).to(DEV).half()
    x = (torch.randn(rows, 384, device=DEV) * 0.7).half()
    res = torch.randn(rows, n_out, device=DEV).half() if residual else None
    assert UA.RS_GEMM and UA.L.lib().devo_upd_rs_supported(n_out, 384)
    outs = {}
    for rs in (True, False):
        UA.RS_GEMM = rs
        try:
            r = res.clone() if residual else None
            outs[rs] = UA._linear_f16(x, lin.weight, lin.bias, relu_from=relu_from, residual=r, out=r).clone()
        finally:
            UA.RS_GEMM = True
    ref = x.double() @ lin.weight.double().t() + lin.bias.double()
    mag = x.double().abs() @ lin.weight.double().abs().t() + lin.bias.double().abs()
    if relu_from is not None:
        ref[:, relu_from:] = ref[:, relu_from:].clamp(min=0)
    if residual:
        ref, mag = ref + res.double(), mag + res.double().abs()
    err = ((outs[True].double() - ref).abs() / mag.clamp(min=1e-6)).max().item()
    assert torch.isfinite(outs[True]).all() and err < 1e-3, err
    assert (outs[True].float() - outs[False].float()).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


def test_update_fp16_operator_row_resident_chain_keeps_its_outputs():
    """The fp16 inference operator with and without the row-resident chain (csrc/gemm_rs.hip: both LayerNorms, both GatedResiduals and the
    heads behind the frame-pair aggregation in one launch; DEVO_UPD_RS_CHAINS): the same outputs within fp16 rounding, at a row count that
    is no multiple of the 96-row tile as well."""
    from devo_amd import update as UA
    from devo_amd import synth
    for n, M, seed in ((15, 96, 5), (7, 13, 6)):
        torch.manual_seed(seed)
        ii, jj, kk = [t.to(DEV) for t in synth.full_graph(n, M)]
        E = ii.numel()
        upd = UA.Update(3).to(DEV).half().eval()
        with torch.no_grad():
            for p in upd.parameters():                                 # (biases and LayerNorm parameters away from their 0 / 1 initial values)
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.1)
        net = torch.randn(1, E, 384, device=DEV).half() * 0.5
        inp = torch.randn(1, E, 384, device=DEV).half() * 0.5
        corr = torch.randn(1, E, 882, device=DEV).half()
        outs = []
        for on in (True, False):
            UA.RS_CHAINS = on
            try:
                with torch.no_grad():
                    o, (d, w, _) = upd(net, inp, corr, None, ii, jj, kk)
            finally:
                UA.RS_CHAINS = True
            outs.append((o.float(), d.float(), w.float()))
        for a, b in zip(*outs):
            assert torch.isfinite(a).all() and (a - b).abs().max().item() <= 1e-2 * max(1.0, b.abs().max().item()), (n, (a - b).abs().max().item())


@pytest.mark.parametrize("rows,n_out", [(18000, 384), (8200, 384)])
def test_row_resident_split_linear_agrees_with_the_ring_kernel(rows, n_out):
    """fp32 storage: csrc/gemm_rs.hip's kernel (rows scaled by their own largest magnitude, split once into two fp16 tiles in LDS, weights
    into registers; DEVO_UPD_RS_SPLIT) and csrc/linear.hip's (running row scales, LDS-DMA rings) compute the same exact-split product:
    both within 1e-6 of the float64 result relative to the magnitude of each output's terms — forward, ReLU from a column on, the in-place
    residual, the ReLU-adjoint gate, and the dX form on the transposed weight."""
    from devo_amd import update as UA
    g = torch.Generator(device="cpu").manual_seed(rows + n_out)
    x = torch.randn(rows, 384, generator=g) * torch.rand(rows, 1, generator=g) * 4
    x[1::7] *= 1e-7
    x[2::11] *= 3e5
    x[3::13, 192:] *= 4096.0
    x[6::19] = 0
    x = x.to(DEV)
    w = (torch.randn(n_out, 384, generator=g) / 384 ** 0.5)
    w[1::5] *= 1e-6
    w[2::9] *= 1e3
    w = w.to(DEV)
    b = torch.randn(n_out, generator=g).to(DEV)
    res = torch.randn(rows, n_out, generator=g).to(DEV)
    gate = torch.randn(rows, n_out, generator=g).to(DEV)
    gy = torch.randn(rows, n_out, generator=g).to(DEV)
    gy[1::3] *= 1e-8
    assert UA.L.lib().devo_upd_rs_split_supported(n_out, 384)
    scale = x.double().abs() @ w.double().abs().t() + b.double().abs()
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    ref_rf = ref.clone(); ref_rf[:, n_out // 2:].relu_()
    for rs in (True, False):
        UA.RS_SPLIT = rs
        try:
            with torch.no_grad():                                      # (the row-resident form serves the inference operator)
                y = UA._linear_split(x, w, b, rs=True)
                y_rf = UA._linear_split(x, w, b, relu_from=n_out // 2, rs=True)
                acc = res.clone()
                y_res = UA._linear_split(x, w, b, residual=acc, out=acc, rs=True)
                y_gate = UA._linear_split(x, w, b, gate=gate, rs=True)
                dx = UA._linear_split(gy, w, None, transposed=True, rs=True) if n_out == 384 else None
        finally:
            UA.RS_SPLIT = True
        assert ((y.double() - ref).abs() / scale).max().item() < 1e-6, rs
        assert ((y_rf.double() - ref_rf).abs() / scale).max().item() < 1e-6, rs
        assert y_res.data_ptr() == acc.data_ptr() and ((y_res.double() - (ref + res.double())).abs() / (scale + res.double().abs())).max().item() < 1e-6, rs
        assert ((y_gate.double() - ref * (gate > 0)).abs() / scale).max().item() < 1e-6, rs
        if dx is not None:
            assert ((dx.double() - gy.double() @ w.double()).abs() / (gy.double().abs() @ w.double().abs() + 1e-300)).max().item() < 1e-6, rs


def test_row_resident_kernels_are_deterministic():
    """csrc/gemm_rs.hip has no atomics: the fp16 operator (three chain kernels + the f | g tails) and the fp32 row-resident layer give the
    same bits on every repetition — a difference would be a missing barrier between the products of a chain (tools/check_rs_determinism.py)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_rs_determinism.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "differences: 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_fp32_operator_under_autocast_takes_the_fp16_kernels():
    """devo.py:311 calls the fp32 update operator under torch.autocast (inference, no gradients): here that call runs the fp16-storage
    operator on a half copy of the parameters — within the fp16 tolerance of the float64 oracle, with autocast's output dtypes (net fp32:
    the reference's LayerNorm / residual stream; delta and weight fp16: Linear outputs), identical to calling a .half() copy directly, the
    copy following the parameters' versions; DEVO_UPD_AUTOCAST_F16=0 keeps the fp32 kernels."""
    from devo_amd import update as UA
    m, sd, net, inp, corr, ii, jj, kk = _random_case()
    ref = U.update(sd, net.double(), inp.double(), corr.double(), ii, jj, kk)
    args = (net.to(DEV), inp.to(DEV), corr.to(DEV), None, ii.to(DEV), jj.to(DEV), kk.to(DEV))
    m = m.to(DEV).eval()
    with torch.no_grad():
        with torch.autocast("cuda", dtype=torch.float16):
            n, (d, w, _) = m(*args)
        assert n.dtype == torch.float32 and d.dtype == torch.float16 and w.dtype == torch.float16
        for got, want, name in ((n, ref[0], "net"), (d, ref[1], "delta"), (w, ref[2], "weight")):
            assert_rel(got.float(), want, 2e-2, f"autocast {name}")
        import copy
        mh = copy.deepcopy(m).half()
        n16, (d16, w16, _) = mh(args[0].half(), args[1].half(), args[2].half(), None, *args[4:])
        # round 6: the first launch reads the fp32 state, the last one writes the new state in fp32 (no conversion passes): a state that IS
        # fp16-exact — what the operator returns and DEVO feeds back — gives the bits of the half copy; any other enters the first sum unrounded
        with torch.autocast("cuda", dtype=torch.float16):
            nr, (dr, wr, _) = m(args[0].half().float(), *args[1:])
        assert torch.equal(n16.float(), nr) and torch.equal(d16, dr) and torch.equal(w16, wr)
        UA.MIXED_STATE = False                                             # ... and with torch's conversions around the fp16 operator (round 5)
        try:
            with torch.autocast("cuda", dtype=torch.float16):
                no, (do, wo, _) = m(*args)
        finally:
            UA.MIXED_STATE = True
        assert no.dtype == torch.float32 and torch.equal(n16.float(), no) and torch.equal(d16, do) and torch.equal(w16, wo)
        assert_rel(n, no, 2e-3, "mixed state against converted state")
        n32, _ = m(*args)                                                  # without autocast: the fp32 kernels
        assert_rel(n32, ref[0], 1e-4, "fp32 net")
        for p in m.parameters():                                           # an optimiser-style step: the half copy follows
            p.mul_(0.5)
        sd2 = {k: v.double() for k, v in m.state_dict().items()}
        ref2 = U.update({k: v.cpu() for k, v in sd2.items()}, net.double(), inp.double(), corr.double(), ii, jj, kk)
        with torch.autocast("cuda", dtype=torch.float16):
            n2, _ = m(*args)
        assert_rel(n2, ref2[0], 2e-2, "autocast net after a parameter update")
        UA.AUTOCAST_F16 = False
        try:
            with torch.autocast("cuda", dtype=torch.float16):
                n3, (d3, _, _) = m(*args)
        finally:
            UA.AUTOCAST_F16 = True
        assert_rel(n3.float(), ref2[0], 1e-3, "fp32 kernels under autocast")


def test_update_operator_on_a_batch_of_sequences():
    """enet.py:80-99 is written for a batch [B, E, dim] sharing one graph; the HIP inference path runs a batch entry by entry: every entry equals the
    single-sequence call (bit for bit) and the torch composition of the same module on the whole batch (1e-4)."""
    from devo_amd import synth
    from devo_amd.update import Update
    ii, jj, kk = [t.to(DEV) for t in synth.full_graph(6, 24)]
    E = ii.numel()
    torch.manual_seed(5)
    upd = Update(3).to(DEV).eval()
    g = torch.Generator().manual_seed(6)
    net = (torch.randn(3, E, 384, generator=g) * 0.3).to(DEV)
    inp = (torch.randn(3, E, 384, generator=g) * 0.3).to(DEV)
    corr = (torch.randn(3, E, 882, generator=g) * 0.3).to(DEV)
    with torch.no_grad():
        n, (d, w, _) = upd(net, inp, corr, None, ii, jj, kk)
        assert n.shape == (3, E, 384) and d.shape == (3, E, 2) and w.shape == (3, E, 2)
        for b in range(3):
            nb, (db, wb, _) = upd(net[b:b + 1], inp[b:b + 1], corr[b:b + 1], None, ii, jj, kk)
            assert torch.equal(n[b:b + 1], nb) and torch.equal(d[b:b + 1], db) and torch.equal(w[b:b + 1], wb)
        nt, (dt_, wt, _) = upd.forward_torch(net, inp, corr, ii, jj, kk)
    assert_rel(n, nt, 1e-4, "batched net")
    assert_rel(d, dt_, 1e-4, "batched delta")
    assert_rel(w, wt, 1e-4, "batched weight")
