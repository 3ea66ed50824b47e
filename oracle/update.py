"""ORACLE (test infrastructure only; never imported by the product path).

CPU restatement of DEVO's `Update` operator — SURVEY.md §8(f) row f1, the caller that sits between the correlation
lookup and the bundle adjustment in every update iteration:

    devo/enet.py:32-99     class Update: corr MLP, LayerNorms, neighbour mixing (c1, c2), two SoftAgg blocks,
                           two GatedResiduals ("gru"), the delta / weight heads
    devo/blocks.py:15-29   GatedResidual:  x + sigmoid(Linear(x)) * (Linear -> ReLU -> Linear)(x)
    devo/blocks.py:31-48   SoftAgg:        h( scatter_sum( f(x) * scatter_softmax(g(x), group), group ) )[group]
    devo/fastba/ba.cpp:104-149  neighbors(kk, jj) (oracle/fastba.py:neighbors)

Functional form: the weights come in as a dict with the reference module's own state_dict keys
("corr.0.weight", "agg_kk.f.bias", "gru.1.gate.0.weight", ...), any width `dim`.
Pinned by tests/golden/update_f64.npz, which tools/gen_golden_update.py produces by running the reference's own
`devo.enet.Update` (imported from /root/reference with shims) on seeded inputs.
"""
import torch
import torch.nn.functional as F
from . import fastba as _fb


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd[key + ".bias"])


def _ln(sd, key, x, eps=1e-3):                               # nn.LayerNorm(dim, eps=1e-3): enet.py:47,53,55,65
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], eps)


def segment_softmax_sum(fx, gx, group):
    """y[b, s, c] = sum over edges e of group s of fx[b,e,c] * softmax over that group of gx[b,e,c]
    (torch_scatter.scatter_softmax + scatter_sum along dim 1, blocks.py:42-43)."""
    n = int(group.max()) + 1 if group.numel() else 0
    B, E, C = fx.shape
    idx = group.view(1, E, 1).expand(B, E, C)
    mx = torch.full((B, n, C), float("-inf"), dtype=fx.dtype).scatter_reduce(1, idx, gx, "amax", include_self=True)
    ex = (gx - mx.gather(1, idx)).exp()
    den = torch.zeros(B, n, C, dtype=fx.dtype).scatter_add(1, idx, ex)
    w = ex / den.gather(1, idx)
    return torch.zeros(B, n, C, dtype=fx.dtype).scatter_add(1, idx, fx * w)


def soft_agg(sd, key, x, ix):
    """SoftAgg.forward with expand=True (blocks.py:39-48)."""
    _, jx = torch.unique(ix, return_inverse=True)
    y = segment_softmax_sum(_lin(sd, key + ".f", x), _lin(sd, key + ".g", x), jx)
    return _lin(sd, key + ".h", y)[:, jx]


def gated_residual(sd, key, x):
    """GatedResidual.forward (blocks.py:28-29)."""
    gate = torch.sigmoid(_lin(sd, key + ".gate.0", x))
    res = _lin(sd, key + ".res.2", torch.relu(_lin(sd, key + ".res.0", x)))
    return x + gate * res


def update(sd, net, inp, corr, ii, jj, kk):
    """Update.forward (enet.py:80-99) -> (net, delta, weight)."""
    c = _lin(sd, "corr.0", corr)
    c = _lin(sd, "corr.2", torch.relu(c))
    c = _lin(sd, "corr.5", torch.relu(_ln(sd, "corr.3", c)))
    net = net + inp + c
    net = _ln(sd, "norm", net)

    ix, jx = _fb.neighbors(kk, jj)                           # enet.py:86
    mask_ix = (ix >= 0).to(net.dtype).reshape(1, -1, 1)
    mask_jx = (jx >= 0).to(net.dtype).reshape(1, -1, 1)
    net = net + _lin(sd, "c1.2", torch.relu(_lin(sd, "c1.0", mask_ix * net[:, ix])))     # net[:, -1] = last row, masked
    net = net + _lin(sd, "c2.2", torch.relu(_lin(sd, "c2.0", mask_jx * net[:, jx])))

    net = net + soft_agg(sd, "agg_kk", net, kk)
    net = net + soft_agg(sd, "agg_ij", net, ii * 12345 + jj)

    net = gated_residual(sd, "gru.1", _ln(sd, "gru.0", net))
    net = gated_residual(sd, "gru.3", _ln(sd, "gru.2", net))

    delta = _lin(sd, "d.1", torch.relu(net))                 # GradientClip is the identity in the forward pass
    weight = torch.sigmoid(_lin(sd, "w.1", torch.relu(net)))
    return net, delta, weight
