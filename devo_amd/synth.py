"""Seeded synthetic inputs for the update + bundle-adjustment hot path.

Everything is drawn from a CPU `torch.Generator` so that the CPU oracle and the HIP path see
bit-identical inputs; callers copy to the device.  Shapes follow the reference:
  poses      [1, n, 7]  world-to-camera, (t, q_xyzw)           (devo/devo.py:152)
  patches    [1, n*M, 3, P, P]  (x, y, inverse depth) at 1/4 resolution (devo/enet.py:190-197)
  intrinsics [1, n, 4]  (fx, fy, cx, cy) / 4                    (devo/enet.py:274)
  fmap       [1, n, C, H, W];  gmap [1, n*M, C, 3, 3]           (devo/enet.py:178-188)
  graph      ii (source frame), jj (target frame), kk (patch)   (devo/enet.py:300-301)
No group exponential is needed: poses are built directly from axis-angle quaternions.
"""
import math
import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def make_poses(n, seed=1234, trans_step=0.05, rot_step=0.01, dtype=torch.float32):
    """Smooth trajectory, pose 0 = identity, unit quaternions."""
    g = _gen(seed)
    axis = torch.randn(n, 3, generator=g, dtype=torch.float64)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    dirs = torch.randn(n, 3, generator=g, dtype=torch.float64)
    k = torch.arange(n, dtype=torch.float64)[:, None]
    ang = rot_step * k
    q = torch.cat([torch.sin(0.5 * ang) * axis, torch.cos(0.5 * ang)], -1)
    t = trans_step * k * dirs / dirs.norm(dim=-1, keepdim=True)
    return torch.cat([t, q], -1).to(dtype)[None]


def make_patches(n, M, H, W, P=3, seed=1234, dtype=torch.float32):
    """P x P pixel grids around integer centres x in [1,W-2], y in [1,H-2] (enet.py:146-147),
    constant inverse depth d ~ U(0.2, 1.0) per patch (enet.py:294-295)."""
    g = _gen(seed + 1)
    x = torch.randint(1, W - 1, (n * M,), generator=g).to(dtype)
    y = torch.randint(1, H - 1, (n * M,), generator=g).to(dtype)
    d = (0.2 + 0.8 * torch.rand(n * M, generator=g, dtype=torch.float64)).to(dtype)
    off = torch.arange(P, dtype=dtype) - P // 2
    gx = x[:, None, None] + off[None, None, :]
    gy = y[:, None, None] + off[None, :, None]
    gx, gy = torch.broadcast_tensors(gx, gy)
    gd = d[:, None, None].expand_as(gx)
    patches = torch.stack([gx, gy, gd], 1)[None].contiguous()
    centres = torch.stack([x, y], -1).view(n, M, 2)
    return patches, centres


def make_intrinsics(n, H, W, dtype=torch.float32):
    """TartanAir 640x480: (320,320,320,240)/4 at H=120,W=160 (tartan.py:189-191); scaled with size."""
    fx = 320.0 / 4 * (W / 160.0)
    return torch.tensor([fx, fx, W / 2.0, H / 2.0], dtype=dtype).expand(1, n, 4).contiguous()


def full_graph(n, M, n_frames=None):
    """enet.py:300-301 style graph: every patch of the first n frames x every frame 0..n-1.
    E = n*n*M; kk-major (each patch's edges are contiguous)."""
    n_frames = n if n_frames is None else n_frames
    kk = torch.arange(n * M).repeat_interleave(n_frames)
    jj = torch.arange(n_frames).repeat(n * M)
    ii = kk // M
    return ii, jj, kk


def sliding_window_graph(n, M, lifetime=13, removal=22, n_first=0):
    """The patch graph DEVO's inference holds after n keyframes (config/default.yaml: PATCH_LIFETIME 13, REMOVAL_WINDOW 22), built the way
    devo/devo.py builds it, frame by frame: for every new keyframe f the forward edges (patches of frames f - 12 .. f - 1 -> frame f,
    devo.py:366-372) and the backward edges (the new frame's patches -> frames f - 12 .. f, :374-380) are appended (:228-231), then the
    edges whose patch lives in a frame older than n - removal are dropped (:304-306), order kept.  Frames [n_first, n).
    Returns ii (source frame = ix[kk]), jj (target frame), kk (global patch index frame * M + m): ~45 k edges at M = 96 once n >= 35; kk is
    NOT sorted and indices grow past mem * M / mem — the lookups see them modulo the ring (devo.py:213-214)."""
    ii, jj, kk = [], [], []
    for f in range(n_first, n):
        nn = f + 1                                               # self.n after the increment of devo.py:537
        t0, t1 = M * max(nn - lifetime, 0), M * max(nn - 1, 0)
        k_f = torch.arange(t0, t1).repeat_interleave(1)           # flatmeshgrid(patches, [f]): every patch once
        kk.append(k_f); jj.append(torch.full_like(k_f, f))
        k_b = torch.arange(M * (nn - 1), M * nn)
        j_b = torch.arange(max(nn - lifetime, 0), nn)
        kk.append(k_b.repeat_interleave(len(j_b))); jj.append(j_b.repeat(len(k_b)))
        kk_all, jj_all = torch.cat(kk), torch.cat(jj)
        keep = (kk_all // M) >= nn - removal
        kk, jj = [kk_all[keep]], [jj_all[keep]]
    kk, jj = kk[0], jj[0]
    return kk // M, jj, kk


def make_features(n, M, C, H, W, centres, seed=1234, dtype=torch.float32):
    """fmap ~ N(0,1)/4 (enet.py:124 '/4.0'); gmap = 3x3 integer patches of fmap at the centres."""
    g = _gen(seed + 2)
    fmap = (torch.randn(n, C, H, W, generator=g) / 4.0)
    cx = centres[..., 0].long()                      # [n,M]
    cy = centres[..., 1].long()
    off = torch.arange(3) - 1
    yy = (cy[:, :, None, None] + off[None, None, :, None]).expand(n, M, 3, 3)
    xx = (cx[:, :, None, None] + off[None, None, None, :]).expand(n, M, 3, 3)
    fr = torch.arange(n)[:, None, None, None].expand(n, M, 3, 3)
    gmap = fmap[fr, :, yy, xx]                        # [n,M,3,3,C]
    gmap = gmap.permute(0, 1, 4, 2, 3).reshape(1, n * M, C, 3, 3).contiguous()
    return fmap[None].to(dtype), gmap.to(dtype)


def pyramid_l1(fmap):
    """avg_pool2d(fmap, 4, 4) (devo/utils.py:70-79)"""
    b, n, c, h, w = fmap.shape
    return torch.nn.functional.avg_pool2d(fmap.float().view(b * n, c, h, w), 4, 4).view(b, n, c, h // 4, w // 4).to(fmap.dtype)


def make_update_outputs(E, seed=1234, sigma=1.0, dtype=torch.float32):
    """Stand-in for the Update operator's outputs: delta ~ N(0, sigma) px, weight ~ U(0,1)
    (sigmoid range, enet.py:73-77)."""
    g = _gen(seed + 3)
    delta = sigma * torch.randn(1, E, 2, generator=g)
    weight = torch.rand(1, E, 2, generator=g)
    return delta.to(dtype), weight.to(dtype)


def workload(name):
    """Named configurations of BASELINE.json."""
    table = {
        "cfg1": dict(n=8, M=48, H=120, W=160, C=128, R=3),       # CPU-runnable case
        "cfg2": dict(n=15, M=96, H=120, W=160, C=128, R=3),      # the metric's configuration
        "cfg2_m80": dict(n=15, M=80, H=120, W=160, C=128, R=3),  # DEVO_base.conf patch count
        "stress": dict(n=32, M=256, H=180, W=320, C=128, R=5),   # HBM-bound altcorr run
        "tiny": dict(n=4, M=6, H=24, W=32, C=16, R=3),
    }
    return dict(table[name])
