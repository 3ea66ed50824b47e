"""Known-answer and cross-check tests that pin the altcorr oracle (the reference has no tests or
golden vectors for altcorr: SURVEY.md §4, §8c)."""
import torch
from oracle import altcorr as A

torch.manual_seed(0)


def _case(n=3, Np=5, C=8, H=9, W=11, E=6, R=2, frac=True, seed=0):
    g = torch.Generator().manual_seed(seed)
    f1 = torch.randn(1, Np, C, 3, 3, generator=g)
    f2 = torch.randn(1, n, C, H, W, generator=g)
    base = torch.stack([torch.rand(E, generator=g) * (W + 4) - 2, torch.rand(E, generator=g) * (H + 4) - 2], 1)
    off = torch.stack(torch.meshgrid(torch.arange(3.) - 1, torch.arange(3.) - 1, indexing="ij")[::-1], 0)  # [2,3,3] x,y
    coords = base[:, :, None, None] + 1.3 * off[None]
    if not frac:
        coords = coords.round()
    ii = torch.randint(0, Np, (E,), generator=g)
    jj = torch.randint(0, n, (E,), generator=g)
    return f1, f2, coords[None].contiguous(), ii, jj, R


def test_vectorised_matches_scalar_loops():
    f1, f2, coords, ii, jj, R = _case()
    a = A.corr_forward(f1, f2, coords, ii, jj, R)
    b = A.corr_forward_scalar(f1, f2, coords, ii, jj, R)
    assert a.shape == (1, 6, 2 * R + 1, 2 * R + 1, 3, 3)
    assert torch.allclose(a, b, atol=1e-12)


def test_integer_coords_are_plain_dot_products():
    """dx = dy = 0  =>  out[b,e,c,a,i0,j0] = <f1[ii[e],:,i0,j0], f2[jj[e],:,y+a-R,x+c-R]> (0 if out of bounds)"""
    f1, f2, coords, ii, jj, R = _case(frac=False, seed=3)
    out = A.corr_forward(f1, f2, coords, ii, jj, R)
    H, W = f2.shape[3:]
    for e in range(coords.shape[1]):
        for i0 in range(3):
            for j0 in range(3):
                x, y = int(coords[0, e, 0, i0, j0]), int(coords[0, e, 1, i0, j0])
                for a in range(2 * R + 1):
                    for c in range(2 * R + 1):
                        yy, xx = y + a - R, x + c - R
                        ref = 0.0
                        if 0 <= yy < H and 0 <= xx < W:
                            ref = float((f1[0, ii[e], :, i0, j0].double() * f2[0, jj[e], :, yy, xx].double()).sum())
                        assert abs(float(out[0, e, c, a, i0, j0]) - ref) < 1e-10


def test_axis_order_is_x_offset_then_y_offset():
    """permute(0,1,3,2,4,5): dim 2 of the result is the x (column) offset.  A feature map that only
    depends on the column must give a result that varies along dim 2 and is constant along dim 3."""
    C, H, W = 4, 12, 12
    f2 = torch.arange(W, dtype=torch.float32).view(1, 1, 1, 1, W).expand(1, 1, C, H, W).contiguous()
    f1 = torch.ones(1, 1, C, 3, 3)
    coords = torch.full((1, 1, 2, 3, 3), 5.0)
    out = A.corr_forward(f1, f2, coords, torch.zeros(1, dtype=torch.long), torch.zeros(1, dtype=torch.long), 2)
    assert torch.allclose(out[0, 0, :, 0, 0, 0], C * (5.0 + torch.arange(-2, 3)).double())
    assert torch.allclose(out[0, 0, 0, :, 0, 0], torch.full((5,), C * 3.0, dtype=torch.float64))


def test_backward_is_adjoint_of_forward():
    f1, f2, coords, ii, jj, R = _case(seed=5)
    f1d = f1.double().requires_grad_(True)
    f2d = f2.double().requires_grad_(True)
    # differentiable re-expression of the forward through torch ops for autograd
    out = A.corr_forward(f1d, f2d, coords, ii, jj, R)
    g = torch.randn(out.shape, dtype=torch.float64)
    (out * g).sum().backward()
    d1, d2 = A.corr_backward(f1.double(), f2.double(), coords, ii, jj, g, R)
    assert torch.allclose(d1, f1d.grad, atol=1e-10)
    assert torch.allclose(d2, f2d.grad, atol=1e-10)


def test_patchify_kat_and_adjoint():
    g = torch.Generator().manual_seed(1)
    net = torch.randn(2, 5, 7, 9, generator=g)
    coords = torch.tensor([[[3.0, 2.0], [0.0, 0.0], [8.7, 6.2]], [[4.5, 3.5], [-3.0, 2.0], [1.0, 6.0]]])
    for R in (0, 1):
        p = A.patchify_forward(net, coords, R)
        D = 2 * R + 2
        assert p.shape == (2, 3, 5, D, D)
        for b in range(2):
            for m in range(3):
                x, y = int(torch.floor(coords[b, m, 0])), int(torch.floor(coords[b, m, 1]))
                for a in range(D):
                    for c in range(D):
                        yy, xx = y + a - R, x + c - R
                        ref = net[b, :, yy, xx] if (0 <= yy < 7 and 0 <= xx < 9) else torch.zeros(5)
                        assert torch.equal(p[b, m, :, a, c], ref)
        gr = torch.randn(p.shape, generator=g)
        back = A.patchify_backward(net, coords, gr, R)
        netd = net.double().requires_grad_(True)
        (A.patchify_forward(netd, coords, R) * gr.double()).sum().backward()
        assert torch.allclose(back.double(), netd.grad, atol=1e-6)
    # wrapper: integer coords => top-left (2R+1)^2 block of the (2R+2)^2 gather (correlation.py:58-66)
    ci = torch.tensor([[[3.0, 2.0], [5.0, 4.0]]])
    assert torch.equal(A.patchify(net[:1], ci, 1), A.patchify_forward(net[:1], ci, 1)[..., :3, :3])
