"""SE3 group type + autograd ops over the HIP kernels — the host-side mirror of the reference's Python
lietorch layer (devo/lietorch/groups.py:51-285 LieGroup/SE3, group_ops.py:7-66 GroupOp and the op
classes, broadcasting.py:9-31), restricted to SE3, the only group the DEVO hot path instantiates.

Same call conventions as the reference: data[..., 7] = (t, q_xyzw); gradients of group elements are
6-vectors in the first 6 of 7 slots; binary ops broadcast leading dimensions; `retr(a) = Exp(a) * X`.
"""
import numpy as np
import torch
from .backends import lietorch_backends as _be

_GID = 3


class _GroupOp(torch.autograd.Function):
    """group_ops.py:7-25: save inputs, call the forward kernel; backward calls the matching *_backward."""
    fwd = bwd = None

    @classmethod
    def forward(cls, ctx, *inputs):
        ctx.save_for_backward(*inputs)
        return cls.fwd(_GID, *inputs)

    @classmethod
    def backward(cls, ctx, grad):
        if cls.bwd is None:
            raise RuntimeError(f"Backward operation not implemented for {cls}")
        return tuple(cls.bwd(_GID, grad.contiguous(), *ctx.saved_tensors))


def _make(name, fwd, bwd):
    return type(name, (_GroupOp,), {"fwd": staticmethod(fwd), "bwd": staticmethod(bwd) if bwd else None})


Exp = _make("Exp", _be.expm, _be.expm_backward)
Log = _make("Log", _be.logm, _be.logm_backward)
Inv = _make("Inv", _be.inv, _be.inv_backward)
Mul = _make("Mul", _be.mul, _be.mul_backward)
Adj = _make("Adj", _be.adj, _be.adj_backward)
AdjT = _make("AdjT", _be.adjT, _be.adjT_backward)
Act3 = _make("Act3", _be.act, _be.act_backward)
Act4 = _make("Act4", _be.act4, _be.act4_backward)
Jinv = _make("Jinv", _be.Jinv, None)
ToMatrix = _make("ToMatrix", _be.as_matrix, None)


def broadcast_inputs(x, y):
    """broadcasting.py:9-31: flatten to [batch, dim] contiguous, broadcasting size-1 leading dims."""
    if y is None:
        return (x.reshape(-1, x.shape[-1]).contiguous(),), tuple(x.shape[:-1])
    if x.dim() != y.dim():
        raise AssertionError("lietorch: operands must have the same number of dimensions")
    xs, ys = x.shape[:-1], y.shape[:-1]
    for n, m in zip(xs, ys):
        if not (n == m or n == 1 or m == 1):
            raise AssertionError(f"lietorch: shapes {tuple(xs)} and {tuple(ys)} do not broadcast")
    out = tuple(max(n, m) for n, m in zip(xs, ys))
    x1 = x.expand(*out, x.shape[-1]).reshape(-1, x.shape[-1]).contiguous()
    y1 = y.expand(*out, y.shape[-1]).reshape(-1, y.shape[-1]).contiguous()
    return (x1, y1), out


def _apply(op, x, y=None):
    inputs, shape = broadcast_inputs(x, y)
    return op.apply(*inputs).view(shape + (-1,))


class SE3:
    """groups.py:266-285 (+ the LieGroup base, :51-233)."""
    group_name = "SE3"
    group_id = _GID
    manifold_dim = 6
    embedded_dim = 7
    id_elem = torch.as_tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0])

    def __init__(self, data):
        self.data = data.data if isinstance(data, SE3) else data

    def __repr__(self):
        return f"{self.group_name}: size={tuple(self.shape)}, device={self.device}, dtype={self.dtype}"

    shape = property(lambda self: self.data.shape[:-1])
    device = property(lambda self: self.data.device)
    dtype = property(lambda self: self.data.dtype)
    tangent_shape = property(lambda self: self.data.shape[:-1] + (self.manifold_dim,))

    # ---- constructors
    @classmethod
    def Identity(cls, *batch_shape, **kwargs):
        if isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        data = cls.id_elem.to(**kwargs).repeat(int(np.prod(batch_shape)), 1)
        return cls(data.view(tuple(batch_shape) + (cls.embedded_dim,)))

    @classmethod
    def IdentityLike(cls, G):
        return cls.Identity(G.shape, device=G.data.device, dtype=G.data.dtype)

    @classmethod
    def Random(cls, *batch_shape, sigma=1.0, **kwargs):
        if isinstance(batch_shape[0], (tuple, list, torch.Size)):
            batch_shape = tuple(batch_shape[0])
        return cls.exp(sigma * torch.randn(tuple(batch_shape) + (cls.manifold_dim,), **kwargs))

    @classmethod
    def exp(cls, a):
        return cls(_apply(Exp, a))

    # ---- group operations
    def log(self):
        return _apply(Log, self.data)

    def inv(self):
        return SE3(_apply(Inv, self.data))

    def mul(self, other):
        return SE3(_apply(Mul, self.data, other.data))

    def retr(self, a):
        return SE3(_apply(Mul, _apply(Exp, a), self.data))

    def adj(self, a):
        return _apply(Adj, self.data, a)

    def adjT(self, a):
        return _apply(AdjT, self.data, a)

    def Jinv(self, a):
        return _apply(Jinv, self.data, a)

    def act(self, p):
        if p.shape[-1] == 3:
            return _apply(Act3, self.data, p)
        if p.shape[-1] == 4:
            return _apply(Act4, self.data, p)
        raise RuntimeError("SE3.act: points must have 3 or 4 components")

    def matrix(self):
        """4x4 matrices, via the action on the identity like the reference (groups.py:180-184)."""
        I = torch.eye(4, dtype=self.dtype, device=self.device).view([1] * (self.data.dim() - 1) + [4, 4])
        return SE3(self.data[..., None, :]).act(I).transpose(-1, -2)

    def translation(self):
        p = torch.eye(4, dtype=self.dtype, device=self.device)[3]               # (made on the device: a list -> GPU tensor is a blocking copy)
        return _apply(Act4, self.data, p.view([1] * (self.data.dim() - 1) + [4]))

    def scale(self, s):
        t, q = self.data.split([3, 4], -1)
        return SE3(torch.cat([t * s.unsqueeze(-1), q], dim=-1))

    def __mul__(self, other):
        if isinstance(other, SE3):
            return self.mul(other)
        if isinstance(other, torch.Tensor):
            return self.act(other)
        return NotImplemented

    # ---- tensor-like plumbing
    def __getitem__(self, index):
        return SE3(self.data[index])

    def __setitem__(self, index, item):
        self.data[index] = item.data

    def detach(self):
        return SE3(self.data.detach())

    def view(self, dims):
        return SE3(self.data.view(tuple(dims) + (self.embedded_dim,)))

    def to(self, *args, **kwargs):
        return SE3(self.data.to(*args, **kwargs))

    def cpu(self):
        return SE3(self.data.cpu())

    def cuda(self):
        return SE3(self.data.cuda())

    def unbind(self, dim=0):
        return [SE3(x) for x in self.data.unbind(dim=dim)]


def cat(group_list, dim):
    return SE3(torch.cat([X.data for X in group_list], dim=dim))


def stack(group_list, dim):
    return SE3(torch.stack([X.data for X in group_list], dim=dim))
