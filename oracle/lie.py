"""ORACLE (test infrastructure only).  Autograd plumbing over oracle/se3.py that mirrors the
reference's Python-side lietorch wrappers (devo/lietorch/group_ops.py:7-66,
groups.py:51-285, broadcasting.py:9-31) closely enough that the oracle restatements of
projective_ops.transform and ba.BA can be differentiated on CPU.

`backend` below is also what tools/gen_golden.py installs as the `lietorch_backends`
shim when it imports the real reference Python modules.
"""
import types
import torch
from . import se3 as K


def _mk_backend():
    """Functions with the pybind signatures of devo/lietorch/src/lietorch.cpp:286-316
    (group_id first; only SE3 = 3 is implemented)."""
    m = types.ModuleType("lietorch_backends")

    def chk(g):
        assert g == 3, "oracle backend implements SE3 (group_id 3) only"

    def un(f):
        def w(g, X):
            chk(g); return f(X)
        return w

    def unb(f):
        def w(g, grad, X):
            chk(g); return [f(grad, X)]
        return w

    def bi(f):
        def w(g, X, y):
            chk(g); return f(X, y)
        return w

    def bib(f):
        def w(g, grad, X, y):
            chk(g); return list(f(grad, X, y))
        return w

    m.expm, m.expm_backward = un(K.expm), unb(K.expm_backward)
    m.logm, m.logm_backward = un(K.logm), unb(K.logm_backward)
    m.inv, m.inv_backward = un(K.inv), unb(K.inv_backward)
    m.mul, m.mul_backward = bi(K.mul), bib(K.mul_backward)
    m.adj, m.adj_backward = bi(K.adj), bib(K.adj_backward)
    m.adjT, m.adjT_backward = bi(K.adjT), bib(K.adjT_backward)
    m.act, m.act_backward = bi(K.act), bib(K.act_backward)
    m.act4, m.act4_backward = bi(K.act4), bib(K.act4_backward)
    m.as_matrix = un(K.as_matrix)
    m.Jinv = bi(K.jinv)

    def projector(g, X):
        raise NotImplementedError("orthogonal_projector is outside the hot path (SURVEY.md §2.1 row 3)")
    m.projector = projector
    return m


backend = _mk_backend()


class _GroupOp(torch.autograd.Function):
    """group_ops.py:7-25"""
    @classmethod
    def forward(cls, ctx, *inputs):
        ctx.save_for_backward(*inputs)
        return cls.fwd(*inputs)

    @classmethod
    def backward(cls, ctx, grad):
        out = cls.bwd(grad.contiguous(), *ctx.saved_tensors)
        return tuple(out) if isinstance(out, (tuple, list)) else (out,)


def _op(fwd, bwd):
    return type("Op", (_GroupOp,), {"fwd": staticmethod(fwd), "bwd": staticmethod(bwd)})


Exp = _op(K.expm, K.expm_backward)
Log = _op(K.logm, K.logm_backward)
Inv = _op(K.inv, K.inv_backward)
Mul = _op(K.mul, K.mul_backward)
AdjT = _op(K.adjT, K.adjT_backward)
Act4 = _op(K.act4, K.act4_backward)


def _bcast(x, y):
    """broadcasting.py:9-31"""
    if y is None:
        return (x.reshape(-1, x.shape[-1]).contiguous(),), x.shape[:-1]
    assert x.dim() == y.dim()
    xs, ys = x.shape[:-1], y.shape[:-1]
    out = tuple(max(n, m) for n, m in zip(xs, ys))
    x1 = x.expand(*out, x.shape[-1]).reshape(-1, x.shape[-1]).contiguous()
    y1 = y.expand(*out, y.shape[-1]).reshape(-1, y.shape[-1]).contiguous()
    return (x1, y1), out


def _apply(op, x, y=None):
    inputs, shp = _bcast(x, y)
    return op.apply(*inputs).view(shp + (-1,))


class SE3:
    """groups.py:266-285 (subset used on the hot path)."""
    def __init__(self, data):
        self.data = data

    @property
    def shape(self):
        return self.data.shape[:-1]

    @classmethod
    def exp(cls, a):
        return cls(_apply(Exp, a))

    def log(self):
        return _apply(Log, self.data)

    def inv(self):
        return SE3(_apply(Inv, self.data))

    def mul(self, o):
        return SE3(_apply(Mul, self.data, o.data))

    def retr(self, a):
        """groups.py:153-156  Exp(a) * X"""
        return SE3(_apply(Mul, _apply(Exp, a), self.data))

    def adjT(self, a):
        return _apply(AdjT, self.data, a)

    def act(self, p):
        assert p.shape[-1] == 4
        return _apply(Act4, self.data, p)

    def matrix(self):
        """groups.py:180-184"""
        I = torch.eye(4, dtype=self.data.dtype).view([1] * (self.data.dim() - 1) + [4, 4])
        return SE3(self.data[..., None, :]).act(I).transpose(-1, -2)

    def __mul__(self, o):
        return self.mul(o) if isinstance(o, SE3) else self.act(o)

    def __getitem__(self, idx):
        return SE3(self.data[idx])

    def detach(self):
        return SE3(self.data.detach())
