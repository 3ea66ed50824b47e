"""Patchifier: the per-frame step in front of the update loop — matching / context features from the two
stride-4 encoders, patch selection, patch gathers, and the correlation pyramid in the lookup kernel's channel-blocked layout.

Mirrors `devo.enet.Patchifier` (enet.py:100-200) with its sub-modules `devo.extractor.BasicEncoder4Evs` (extractor.py:269-335,
residual blocks :6-54) and `devo.selector.Scorer` / `PatchSelector` (selector.py:19-47, 50-260): the parameter tree has the
reference's names and shapes (a reference checkpoint loads with `load_state_dict`), forward returns the reference's tuple.

What is different, on purpose (MI355X):
  * the convolutions are the only dense contractions of the path: they go to MIOpen through PyTorch-ROCm, in
    channels-last memory format (MIOpen's native NHWC kernels on gfx950; under `torch.autocast(fp16 | bf16)` on the matrix
    cores).  Nothing here is a hand-written kernel, by SURVEY's own scoping of this row;
  * the three gathers are the HIP `patchify` kernel (devo_amd.altcorr.patchify); the `[x, y, inverse depth]` patches are
    written in closed form instead of gathering a materialised coordinate grid (utils.py:38-59);
  * `pyramid()` builds both pyramid levels straight in the channel-blocked layout of the lookup kernel (devo_pyramid_build)
    instead of NCHW + avg_pool2d (devo.py:526-527).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import altcorr

# MIOpen "finds" a solver for every new convolution configuration by compiling and timing all candidates — on a fresh box 34-121 s
# for this module's forward + backward at DEVO's input size (121 s of it in the naive weight-gradient reference kernels that take
# part in the contest; tools/miopen_first_call.py).  devo_amd/miopen_db/ carries the find results (text) and the compiled kernels
# of this image's MIOpen for the configurations bench.py and the tools use: with it the first call costs 0.4 s.  Only defaults:
# a caller's own MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR win, and without the directory MIOpen behaves as usual.
_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
if os.path.isdir(_DB):
    os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(_DB, "config"))
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(_DB, "cache"))

DIM_INET, DIM_FNET, DIM_ENC = 384, 128, 32
_GRAPH = os.environ.get("DEVO_PATCHIFIER_GRAPH", "1") != "0"      # 0: the encoders' launches one by one, every call
_GRAPH_MAX_FRAMES = 2                                              # (beyond a frame or two the call is paced by the GPU: nothing to gain)
_LOWP_CL = os.environ.get("DEVO_PATCHIFIER_LOWP_CL", "0") == "1"   # (experiment: the copy's weights in channels-last format)
_LOWP = os.environ.get("DEVO_PATCHIFIER_LOWP", "1") != "0"        # 0: an autocast call converts the parameters itself, every call


def _norm(kind, x):
    if kind == "instance":
        return F.instance_norm(x)                     # nn.InstanceNorm2d defaults: no affine, no running statistics
    if kind == "none":
        return x
    raise NotImplementedError(f"norm_fn = {kind!r} (DEVO uses 'instance' for fnet and 'none' for inet)")


_FUSED_IN = os.environ.get("DEVO_PATCHIFIER_FUSED_NORM", "1") != "0"   # 0: F.instance_norm + F.relu (+ the sum) as separate ATen kernels, always


def _fusable(x, *others):
    if not (_FUSED_IN and not torch.is_grad_enabled() and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float16, torch.float32)
            and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % (8 if x.dtype == torch.float16 else 4) == 0 and x.data_ptr() % 16 == 0):
        return False
    for o in others:
        if o is not None and not (o.dtype == x.dtype and o.data_ptr() % 16 == 0 and (o.dim() == 1 or (o.shape == x.shape and o.is_contiguous(memory_format=torch.channels_last)))):
            return False
    return True


def _conv_nobias(m, x):
    """the convolution of module `m` without its bias (the fused kernels behind it add the bias as ATen would: one elementwise launch less)"""
    return F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups)


def _bias_act(x, relu=True, residual=None, bias=None):
    """act(x + bias) — with `residual`: relu(residual + relu(x + bias)) — as one launch on a channels-last activation (devo_bias_act_cl)."""
    if bias is not None and bias.dtype != x.dtype:
        bias = bias.to(x.dtype)                                                   # (fp32 parameters under autocast: the convolution would have cast it too)
    if _fusable(x, bias, residual) and (residual is None or relu):
        from . import _lib as L
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        L.check(L.lib().devo_bias_act_cl(L.ptr(x), L.ptr(bias), L.ptr(residual), L.ptr(y), N * H * W, C, int(bool(relu)), L.dtype_code(x), L.stream()),
                "patchifier.bias_act")
        return y
    y = x if bias is None else x + bias.view(1, -1, 1, 1)
    if relu:
        y = F.relu(y)
    return F.relu(residual + y) if residual is not None else y


def _in_relu(x, relu=True, residual=None, bias=None):
    """relu(instance_norm(x)) — or, with `residual`, relu(residual + relu(instance_norm(x))): the tail of a residual block — in two launches on a
    channels-last activation when no gradient is needed (devo_instnorm_cl; through ATen one norm of an NHWC tensor is a copy to NCHW, a statistics
    kernel, a transform kernel and the ReLU: 10 norms per frame were half of the encoders' GPU time).  Anything else: the ATen composition."""
    if bias is not None and bias.dtype != x.dtype:
        bias = bias.to(x.dtype)
    if _fusable(x, bias, residual) and (residual is None or relu):
        from . import _lib as L
        N, C, H, W = x.shape
        y = torch.empty_like(x)                                                   # (keeps the channels-last strides)
        ws = torch.empty(int(L.lib().devo_instnorm_workspace_bytes(N, C)), dtype=torch.uint8, device=x.device)
        L.check(L.lib().devo_instnorm_bias_cl(L.ptr(x), L.ptr(bias), L.ptr(residual), L.ptr(y), N, H * W, C, 1e-5, int(bool(relu)), L.ptr(ws), ws.numel(),
                                              L.dtype_code(x), L.stream()), "patchifier.instance_norm")
        return y
    if bias is not None:
        x = x + bias.view(1, -1, 1, 1)
    y = F.instance_norm(x)
    if relu:
        y = F.relu(y)
    return F.relu(residual + y) if residual is not None else y


class _Residual(nn.Module):
    """extractor.py:6-54: two 3x3 convolutions + identity (or a strided 1x1 projection), ReLU after the sum."""

    def __init__(self, cin, cout, norm_fn, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.norm_fn = norm_fn
        # same child names as the reference, so that state-dict keys line up (the norms hold no parameters)
        self.downsample = None if stride == 1 else nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride))

    def forward(self, x):
        if not torch.is_grad_enabled() and _FUSED_IN and x.is_cuda:              # inference: bias, norm, ReLU (+ the block's sum and ReLU) fused
            post = _in_relu if self.norm_fn == "instance" else _bias_act
            y = post(_conv_nobias(self.conv1, x), bias=self.conv1.bias)
            if self.downsample is not None:
                d = self.downsample[0]
                x = post(_conv_nobias(d, x), relu=False, bias=d.bias)
            return post(_conv_nobias(self.conv2, y), residual=x, bias=self.conv2.bias)
        y = F.relu(_norm(self.norm_fn, self.conv1(x)))
        y = F.relu(_norm(self.norm_fn, self.conv2(y)))
        if self.downsample is not None:
            x = _norm(self.norm_fn, self.downsample(x))
        return F.relu(x + y)


def _init(module):
    for m in module.modules():                        # extractor.py:296-303 / selector.py:33-40
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class Encoder(nn.Module):
    """`BasicEncoder4Evs` (extractor.py:269-335): 7x7/2 stem, two residual stages (dim, 2 dim; the second strided), 1x1 head.
    [b, n, bins, H, W] -> [b, n, output_dim, H/4, W/4]."""

    def __init__(self, bins=5, output_dim=128, dim=DIM_ENC, norm_fn="instance"):
        super().__init__()
        if norm_fn not in ("instance", "none"):
            raise NotImplementedError(f"norm_fn = {norm_fn!r}")
        self.norm_fn = norm_fn
        self.conv1 = nn.Conv2d(bins, dim, 7, stride=2, padding=3)
        self.layer1 = nn.Sequential(_Residual(dim, dim, norm_fn, 1), _Residual(dim, dim, norm_fn, 1))
        self.layer2 = nn.Sequential(_Residual(dim, 2 * dim, norm_fn, 2), _Residual(2 * dim, 2 * dim, norm_fn, 1))
        self.conv2 = nn.Conv2d(2 * dim, output_dim, 1)
        _init(self)

    def forward(self, x):
        b, n = x.shape[:2]
        x = x.reshape(b * n, *x.shape[2:]).contiguous(memory_format=torch.channels_last)
        if not torch.is_grad_enabled() and _FUSED_IN and x.is_cuda:
            post = _in_relu if self.norm_fn == "instance" else _bias_act
            x = post(_conv_nobias(self.conv1, x), bias=self.conv1.bias)
            x = _bias_act(_conv_nobias(self.conv2, self.layer2(self.layer1(x))), relu=False, bias=self.conv2.bias)
        else:
            x = F.relu(_norm(self.norm_fn, self.conv1(x)))
            x = self.conv2(self.layer2(self.layer1(x)))
        return x.reshape(b, n, *x.shape[1:])


class Scorer(nn.Module):
    """selector.py:19-47: four unpadded 3x3 convolutions (bins -> 8 -> 16 -> 32 -> 1) and a 4x4 max-pool: one score per
    stride-4 cell, [b, n, (H - 8) // 4, (W - 8) // 4]."""

    def __init__(self, bins=5):
        super().__init__()
        self.scorer = nn.Sequential(nn.Conv2d(bins, 8, 3), nn.ReLU(inplace=True), nn.Conv2d(8, 16, 3), nn.ReLU(inplace=True),
                                    nn.Conv2d(16, 32, 3), nn.ReLU(inplace=True), nn.Conv2d(32, 1, 3), nn.MaxPool2d(4, 4))
        _init(self)

    def forward(self, x):
        b, n = x.shape[:2]
        x = x.reshape(b * n, *x.shape[2:]).contiguous(memory_format=torch.channels_last)
        if not torch.is_grad_enabled() and _FUSED_IN and x.is_cuda:              # inference: bias + ReLU behind a bias-free convolution, one launch
            for i in (0, 2, 4):
                x = _bias_act(_conv_nobias(self.scorer[i], x), bias=self.scorer[i].bias)
            s = self.scorer[7](self.scorer[6](x))
        else:
            s = self.scorer(x)
        return s.reshape(b, n, *s.shape[2:])


# ---------------------------------------------------------------------------------------------------------------- selection
def _quadrant_order(idx, h2, w2):
    """cell indices chosen inside each of the 2 x 2 quadrants (idx [bn, k, 4], quadrant-local, row-major over h2 x w2)
    -> x, y on the whole pooled map, [bn, 4 k] in the reference's (k-major, quadrant-minor) order (selector.py:72-93)."""
    x, y = idx % w2, torch.div(idx, w2, rounding_mode="floor")
    # (made on the device: torch.tensor(list, device=cuda) is a blocking host -> device copy — it waited for everything enqueued before it, the
    #  encoders of this frame included: 1.1 ms of host time per frame in 'multi' / 'topk' selection, and no stream capture through it)
    q = torch.arange(4, device=idx.device)
    qx = (q % 2) * w2
    qy = torch.div(q, 2, rounding_mode="floor") * h2
    return (x + qx).flatten(1), (y + qy).flatten(1)


def _quadrants(pooled):
    """[bn, h1, w1] -> [bn, 4, h2 * w2] (the four quadrants of the top-left 2 h2 x 2 w2 part, selector.py:59-70)"""
    bn, h1, w1 = pooled.shape
    h2, w2 = h1 // 2, w1 // 2
    q = pooled[:, :2 * h2, :2 * w2].reshape(bn, 2, h2, 2, w2).permute(0, 1, 3, 2, 4)
    return q.reshape(bn, 4, h2 * w2), h2, w2


def select_three_x_random(scores, m, candidates=None):
    """3 m uniform candidates per frame, the m with the highest score win (selector.py:95-108 / enet.py:146-158).
    scores [1, n, h, w] -> x + 1, y + 1 (feature-map pixels), the winners' scores in ascending order."""
    _, n, h, w = scores.shape
    if candidates is None:
        x = torch.randint(0, w, (n, 3 * m), device=scores.device)
        y = torch.randint(0, h, (n, 3 * m), device=scores.device)
    else:
        x, y = candidates
    s = scores[0][torch.arange(n, device=scores.device)[:, None], y, x]          # = patchify(scores, coords, 0) at integer coords
    vs, ix = torch.sort(s, dim=1)
    top = ix[:, -m:]
    return torch.gather(x, 1, top) + 1, torch.gather(y, 1, top) + 1, vs[:, -m:].contiguous()


def select_topk(scores, m, grid=True, k=4):
    """pooled top-k (selector.py:152-192): best pixel of every k x k cell, then the m best cells (m / 4 per quadrant with
    `grid`)."""
    b, n, h, w = scores.shape
    cells = F.unfold(scores.reshape(b * n, 1, h, w), kernel_size=k, stride=k)    # [bn, k*k, cells]
    best, off = cells.max(dim=1)
    h1, w1 = h // k, w // k
    if grid:
        q, h2, w2 = _quadrants(best.reshape(b * n, h1, w1))
        idx = torch.topk(q, m // 4, dim=-1).indices.transpose(1, 2)              # [bn, m/4, 4]
        cx, cy = _quadrant_order(idx, h2, w2)
    else:
        idx = torch.topk(best, m, dim=-1).indices
        cx, cy = idx % w1, torch.div(idx, w1, rounding_mode="floor")
    o = torch.gather(off, 1, cy * w1 + cx)
    return k * cx + o % k, k * cy + torch.div(o, k, rounding_mode="floor")


def select_multi(scores, m, grid=True, k=4):
    """average-pooled multinomial sampling (selector.py:110-150): cells drawn with probability ~ their mean score (m / 4 per
    quadrant with `grid`), then one pixel of the cell's k x k window (the reference's windows start one pixel up-left of
    the cell: unfold(padding = 1)) with probability ~ its score."""
    b, n, h, w = scores.shape
    avg = F.avg_pool2d(scores, k, k).reshape(b * n, h // k, w // k)
    h1, w1 = avg.shape[1:]
    if grid:
        q, h2, w2 = _quadrants(avg)
        idx = torch.multinomial(q.reshape(b * n * 4, h2 * w2) + 1e-7, m // 4).reshape(b * n, 4, m // 4).transpose(1, 2)
        cx, cy = _quadrant_order(idx, h2, w2)
    else:
        idx = torch.multinomial(avg.reshape(b * n, -1), m)
        cx, cy = idx % w1, torch.div(idx, w1, rounding_mode="floor")
    win = F.unfold(scores.reshape(b * n, 1, h, w), kernel_size=k, stride=k, padding=1).transpose(1, 2)   # [bn, windows, k*k]
    pick = torch.gather(win, 1, (cy * w1 + cx)[..., None].expand(-1, -1, k * k)) + 1e-7
    o = torch.multinomial(pick.flatten(0, 1), 1).reshape(b * n, m)
    return k * cx + o % k, k * cy + torch.div(o, k, rounding_mode="floor")


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms restated (this image has no torchvision): greedy non-maximum suppression inside every category —
    boxes [N, 4] (x1, y1, x2, y2), in decreasing order of score a box is kept unless a kept box of its category overlaps it with
    IoU > threshold — and the kept indices in decreasing order of score.  Per category the greedy recursion
    keep[j] = not any_{i < j} (keep[i] and IoU[i, j] > thr) is iterated as a whole-vector fixpoint (element j is final once all i < j
    are: at most as many rounds as the longest suppression chain)."""
    if boxes.numel() == 0:
        return torch.empty(0, dtype=torch.long, device=boxes.device)
    order = torch.argsort(scores, descending=True, stable=True)
    b, c = boxes[order].float(), idxs[order]
    keep = torch.ones(order.numel(), dtype=torch.bool, device=boxes.device)
    for cat in torch.unique(c).tolist():
        sel = (c == cat).nonzero()[:, 0]
        bb = b[sel]
        area = (bb[:, 2] - bb[:, 0]) * (bb[:, 3] - bb[:, 1])
        iw = (torch.minimum(bb[:, None, 2], bb[None, :, 2]) - torch.maximum(bb[:, None, 0], bb[None, :, 0])).clamp(min=0)
        ih = (torch.minimum(bb[:, None, 3], bb[None, :, 3]) - torch.maximum(bb[:, None, 1], bb[None, :, 1])).clamp(min=0)
        inter = iw * ih
        over = torch.triu(inter / (area[:, None] + area[None, :] - inter) > iou_threshold, diagonal=1)      # [i, j]: i (higher score) can suppress j
        kc = torch.ones(sel.numel(), dtype=torch.bool, device=boxes.device)
        while True:
            new = ~(over & kc[:, None]).any(0)
            if torch.equal(new, kc):
                break
            kc = new
        keep[sel] = kc
    return order[keep]


def select_nms(scores, m, grid=True, k=4, radius=1.5, iou=0.4):
    """pooled NMS sampling (selector.py:194-254): the best pixel of every k x k cell, a (2 radius)^2 box at it (clamped at the top / left
    border like the reference's), non-maximum suppression among the boxes of a frame — of a frame's quadrant with `grid`, the quadrant
    test written as the reference writes it (the box corner in PIXELS against half the POOLED size) — and the m best survivors per frame.
    A frame with fewer than m survivors is an error (the reference fails in its torch.cat at that point, selector.py:250-252)."""
    b, n, h, w = scores.shape
    best, at = F.max_pool2d(scores, kernel_size=k, stride=k, return_indices=True)
    h1, w1 = best.shape[-2:]
    at = at.reshape(b * n, h1 * w1)
    cx, cy = at % w, torch.div(at, w, rounding_mode="floor")
    x1, y1 = (cx.float() - radius).clamp(min=0.0), (cy.float() - radius).clamp(min=0.0)
    boxes = torch.stack([x1, y1, x1 + 2 * radius, y1 + 2 * radius], dim=-1).reshape(-1, 4)
    frame = torch.arange(b * n, device=scores.device)[:, None].expand(-1, h1 * w1).reshape(-1)
    cat = frame
    if grid:
        left, up = x1.reshape(-1) < w1 / 2, y1.reshape(-1) < h1 / 2
        cat = 4 * frame + (~left).long() + 2 * (~up).long()
    kept = batched_nms(boxes, best.reshape(-1), cat, iou)
    fk, xk, yk = frame[kept], cx.reshape(-1)[kept], cy.reshape(-1)[kept]
    xs, ys = [], []
    for f in range(b * n):
        sel = (fk == f).nonzero()[:m, 0]
        if sel.numel() < m:
            raise RuntimeError(f"patch selection 'nms': frame {f} keeps {sel.numel()} of the {m} patches asked for")
        xs.append(xk[sel]); ys.append(yk[sel])
    return torch.stack(xs), torch.stack(ys)


def select(scores, m, mode, grid=True, k=4):
    """PatchSelector.__call__ (selector.py:256-287): the score map is zero-padded (centred) to whole cells — whole 2 x 2 grids of
    cells with `grid` —, the method runs on the padded map, the coordinates are shifted back and clamped into the map."""
    mode = mode.lower()
    if mode not in ("3xrandom", "topk", "multi", "nms"):
        raise NotImplementedError(f"patch selection mode {mode!r} (have: 3xrandom, topk, multi, nms)")
    h, w = scores.shape[-2:]
    f = 2 * k if grid else k
    ph, pw = (f - h % f) % f, (f - w % f) % f
    top, left = ph // 2, pw // 2
    padded = F.pad(scores, (left, pw - left, top, ph - top))
    if mode == "3xrandom":                       # candidates are drawn on the padded map like every other method (selector.py:92-105,266-286)
        x, y = select_three_x_random(padded, m)[:2]
    else:
        x, y = {"topk": select_topk, "multi": select_multi, "nms": select_nms}[mode](padded, m, grid, k)
    return (x - left).clamp(min=0, max=w - 1), (y - top).clamp(min=0, max=h - 1)


# --------------------------------------------------------------------------------------------------------------- the module
class Patchifier(nn.Module):
    """enet.py:100-200.  forward(images [1, n, bins, H, W]) -> fmap [1, n, 128, H/4, W/4], gmap [1, n M, 128, P, P],
    imap [1, n M, 384, 1, 1], patches [1, n M, 3, P, P] (x, y, inverse depth), index [n M] (+ scores when training with the
    scorer, + colour when asked for in eval)."""

    def __init__(self, patch_size=3, dim_inet=DIM_INET, dim_fnet=DIM_FNET, dim=DIM_ENC, patch_selector="scorer", bins=5):
        super().__init__()
        self.patch_size, self.dim_inet, self.dim_fnet = patch_size, dim_inet, dim_fnet
        self.patch_selector = patch_selector.lower()
        if self.patch_selector not in ("scorer", "gradient", "random"):
            raise NotImplementedError(f"patch_selector = {patch_selector!r}")
        self.fnet = Encoder(bins, dim_fnet, dim, "instance")
        self.inet = Encoder(bins, dim_inet, dim, "none")
        if self.patch_selector == "scorer":
            self.scorer = Scorer(bins)

    # ---- inference under autocast: the encoders on a low-precision copy of their parameters (round 6).  `devo.py:250` calls patchify under
    # `torch.autocast` with fp32 parameters once per frame: autocast then converts every convolution's weight and bias again for every call (its
    # cast cache lives as long as the autocast context: one frame) — 70 launches and 0.24 ms of GPU time of a call that is paced by its ~300
    # launches (2.35 ms per frame of 480 x 640).  The copy (fp16 / bf16, channels-last weights) is kept per version of the parameters; the
    # arithmetic is autocast's own (convolutions, instance norms, ReLUs and sums in the low precision), the outputs are the same tensors.
    def _lowp_modules(self, dtype, cl=False):
        plist = self.__dict__.get("_plist")
        if plist is None or len(plist[1]) != plist[0]:
            ps = list(self.parameters())
            plist = self.__dict__["_plist"] = (len(ps), ps)                      # (walking the module tree costs 0.2 ms per call)
        key = (dtype, bool(cl), tuple((p.data_ptr(), p._version) for p in plist[1]))
        slot = "_lowp_cl" if cl else "_lowp"
        sh = self.__dict__.get(slot)
        if sh is None or sh[0] != key:
            import copy
            mods = {}
            for name in ("fnet", "inet", "scorer"):
                m = getattr(self, name, None)
                if m is None:
                    continue
                c = copy.deepcopy(m).to(dtype).eval()
                if cl or _LOWP_CL:                                             # (channels-last weights: no per-call re-layout kernel in front of every
                    for mod in c.modules():                                      #  convolution — 4.7 us each; with many frames MIOpen picks slower kernels for them)
                        if isinstance(mod, nn.Conv2d):
                            mod.weight.data = mod.weight.data.contiguous(memory_format=torch.channels_last)
                for q in c.parameters():
                    q.requires_grad_(False)
                mods[name] = c
            sh = (key, mods)
            self.__dict__[slot] = sh
        return sh[1]

    # One frame at a time (devo.py:250) the three CNNs are ~110 launches of a few microseconds each and the call is paced by the host (2.0 ms per
    # frame for 1.5 ms of GPU work).  From the third call with the same input shape on, the encoders + scorer replay from ONE HIP graph (static
    # input: the low-precision copy of the images; the outputs are handed out as copies: nothing the caller holds is overwritten by the next frame).
    # Never while the caller captures a graph of its own; DEVO_PATCHIFIER_GRAPH=0 switches it off.
    def _encode_lowp(self, images, lowp, dtype, with_scorer):
        def run(x):
            fm = lowp["fnet"](x) / 4.0
            im = lowp["inet"](x) / 4.0
            sm = torch.sigmoid(lowp["scorer"](x).float()) if with_scorer else None
            return fm, im, sm
        frames = images.shape[0] * images.shape[1]
        with torch.autocast("cuda", enabled=False):
            if _GRAPH and frames <= _GRAPH_MAX_FRAMES and not torch.cuda.is_current_stream_capturing():
                key = (tuple(images.shape), images.dtype, dtype, bool(with_scorer), str(images.device))
                st = self.__dict__.get("_enc_graph")
                if st is None or st["key"] != key or st["lowp"] is not lowp:     # (the graph holds the parameter copy it was captured on alive)
                    st = {"key": key, "lowp": lowp, "calls": 0, "graph": None}
                    self.__dict__["_enc_graph"] = st
                st["calls"] += 1
                if st["graph"] is None and st["calls"] >= 3:                    # (two eager calls first: MIOpen has picked its kernels by then)
                    cur = torch.cuda.current_stream()
                    x_static = torch.empty(images.shape, dtype=dtype, device=images.device)
                    side = torch.cuda.Stream(device=images.device)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        x_static.copy_(images)
                        run(x_static)
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, stream=side):
                            outs = run(x_static)
                    cur.wait_stream(side)
                    st.update(graph=g, x=x_static, outs=outs)
                if st["graph"] is not None:
                    st["x"].copy_(images)
                    st["graph"].replay()
                    return tuple(o.clone() if o is not None else None for o in st["outs"])
            return run(images.to(dtype))

    def __getstate__(self):
        d = self.__dict__.copy()                                               # (copy.deepcopy / torch.save: the per-version copy stays behind)
        d.pop("_lowp", None)
        d.pop("_lowp_cl", None)
        d.pop("_enc_graph", None)
        d.pop("_plist", None)
        return d

    def train(self, mode=True):
        self.__dict__.pop("_plist", None)                                      # (mode switches are where checkpoints / replaced parameters come in)
        return super().train(mode)

    def _apply(self, fn, recurse=True):
        self.__dict__.pop("_lowp", None)                                       # .to() / .half() / .cuda(): the parameters' storage moves
        self.__dict__.pop("_lowp_cl", None)
        self.__dict__.pop("_enc_graph", None)
        self.__dict__.pop("_plist", None)
        return super()._apply(fn, recurse)

    @staticmethod
    def event_gradient(images):
        """enet.py:112-118: gradient magnitude of the event count image at stride 4."""
        s = images.sum(dim=2)
        dx = s[..., :-1, 1:] - s[..., :-1, :-1]
        dy = s[..., 1:, :-1] - s[..., :-1, :-1]
        return F.avg_pool2d(torch.sqrt(dx * dx + dy * dy), 4, 4)

    def forward(self, images, patches_per_image=80, disps=None, return_color=False, scorer_eval_mode="multi",
                scorer_eval_use_grid=True, candidates=None, coords=None):
        """`candidates` = (x, y) int64 [n, 3 M]: the uniform draws of the training branch, `coords` = (x, y) [n, M]: the final
        patch centres — both optional, for reproducible tests (the reference draws them on the device)."""
        lowp = None
        smap_lp = None
        if (_LOWP and not torch.is_grad_enabled() and images.is_cuda and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") in (torch.float16, torch.bfloat16) and self.fnet.conv1.weight.dtype == torch.float32):
            lowp = self._lowp_modules(torch.get_autocast_dtype("cuda"), cl=images.shape[0] * images.shape[1] <= _GRAPH_MAX_FRAMES)
            fmap, imap, smap_lp = self._encode_lowp(images, lowp, torch.get_autocast_dtype("cuda"),
                                                   coords is None and self.patch_selector == "scorer")
        else:
            fmap = self.fnet(images) / 4.0
            imap = self.inet(images) / 4.0
        b, n, _, h, w = fmap.shape
        P, M, dev = self.patch_size, patches_per_image, fmap.device
        scores = None
        if coords is not None:
            x, y = coords
        elif self.patch_selector == "gradient":
            g = self.event_gradient(images)
            x, y = select(g, M, "3xrandom" if self.training else scorer_eval_mode, scorer_eval_use_grid) if candidates is None else \
                select_three_x_random(g, M, candidates)[:2]
            x, y = x.clamp(min=1, max=w - 2), y.clamp(min=1, max=h - 2)
        elif self.patch_selector == "random":
            x = torch.randint(1, w - 1, (n, M), device=dev)
            y = torch.randint(1, h - 1, (n, M), device=dev)
        else:
            if smap_lp is not None:
                smap = smap_lp
            else:
                smap = torch.sigmoid(self.scorer(images).float())               # [1, n, h - 2, w - 2]
            if self.training:
                x, y, scores = select_three_x_random(smap, M, candidates)
            else:
                x, y = select(smap, M, scorer_eval_mode, scorer_eval_use_grid)
                scores = smap[0][torch.arange(n, device=dev)[:, None], y, x]
                x, y = x + 1, y + 1
        if self.patch_selector == "scorer" and coords is not None:
            smap = torch.sigmoid(self.scorer(images).float())
            scores = smap[0][torch.arange(n, device=dev)[:, None], (y - 1).clamp(0, h - 3), (x - 1).clamp(0, w - 3)]
        xy = torch.stack([x, y], dim=-1).float()                                  # [n, M, 2], feature-map pixels
        # (the kernel takes the channels-last strides: no NCHW copy; without gradients the gathers read the maps in their own precision and only the
        #  gathered patches are widened — the same values as widening 9.8 M map elements first)
        lazy = not torch.is_grad_enabled() and fmap.dtype in (torch.float16, torch.float32) and imap.dtype == fmap.dtype
        src_i, src_f = (imap[0], fmap[0]) if lazy else (imap[0].float(), fmap[0].float())
        imap_p = altcorr.patchify(src_i, xy, 0).float().view(b, -1, self.dim_inet, 1, 1)
        gmap = altcorr.patchify(src_f, xy, P // 2).float().view(b, -1, self.dim_fnet, P, P)
        # patches = patchify(coords_grid_with_index(disps), xy, P // 2) in closed form: pixel (x + j - r, y + i - r) and its depth
        r = P // 2
        off = torch.arange(-r, r + 1, device=dev, dtype=torch.float32)
        px = (xy[..., 0, None, None] + off[None, None, None, :]).expand(n, M, P, P)
        py = (xy[..., 1, None, None] + off[None, None, :, None]).expand(n, M, P, P)
        if disps is None:
            pd = torch.ones(n, M, P, P, device=dev)
        else:
            pd = altcorr.patchify(disps[0, :, None].float().contiguous(), xy, r).view(n, M, P, P)
            inside = (px >= 0) & (px < w) & (py >= 0) & (py < h)                  # (the gather returns 0 outside the frame; so does the grid's)
            px, py = px * inside, py * inside
        patches = torch.stack([px, py, pd], dim=2).view(b, n * M, 3, P, P)
        index = torch.arange(n, device=dev).view(n, 1).repeat(1, M).reshape(-1)
        if self.training and self.patch_selector == "scorer":
            return fmap, gmap, imap_p, patches, index, scores
        if not self.training and return_color:
            clr = altcorr.patchify(images[0].abs().sum(dim=1, keepdim=True).float().contiguous(), 4 * (xy + 0.5), 0).clamp(min=0, max=255).view(b, -1, 1)
            return fmap, gmap, imap_p, patches, index, clr
        return fmap, gmap, imap_p, patches, index

    @staticmethod
    def pyramid(fmap, dtype=None):
        """the two lookup levels of `fmap` [1, n, C, h, w] (level 1 = 4x4 mean, devo.py:526-527) in the lookup kernel's
        channel-blocked layout, one pass (devo_pyramid_build)."""
        f = fmap if dtype is None else fmap.to(dtype)
        return altcorr.build_pyramid(f.contiguous())

    def num_parameters(self):
        return sum(p.numel() for p in self.parameters())
