"""Training step of the update + bundle-adjustment path under data parallelism (BASELINE.json configurations 3 and 4).

What the reference trains (train.py:31-42, 90-107, 172-246; devo/enet.py:203-370): one process per GPU, `DDP(net)`, every
rank draws its own sequences (DistributedSampler), runs the recurrent update for STEPS iterations — reprojection, 2-level
correlation lookup (gradients through 20 % of the edges, correlation.py:20-25), Update operator, two differentiable
Gauss-Newton steps — accumulates flow + pose losses, and `loss.backward()` all-reduces ONE bucket of 3 397 061 fp32
gradients (13.59 MB: Update 3 004 804 + fnet 184 576 + inet 201 216 + scorer 6 465) over NCCL = RCCL.

Here the same step runs on this repo's path: devo_amd.patchifier (the two encoders + scorer through MIOpen, HIP gathers),
projective_ops / altcorr / update / ba (HIP kernels + autograd) on synthetic TartanAir-shaped inputs (random voxel
grids, the synthetic trajectory's poses and patch centres).  The module tree is the reference's (`update`, `patchify.fnet`,
`patchify.inet`, `patchify.scorer`): the gradient bucket DDP all-reduces is the reference's by construction.
"""
import torch
import torch.nn as nn

from . import synth

N_FNET, N_INET, N_SCORER = 184_576, 201_216, 6_465          # parameter counts of the modules this path does not build
N_TOTAL = 3_397_061
FUSED_LOOKUP = __import__("os").environ.get("DEVO_TRAIN_FUSED_LOOKUP", "1") != "0"     # 0: two altcorr.corr calls + torch.stack, as enet.py:203-216 writes it


class TrainNet(nn.Module):
    """The parameters the reference's DDP wraps (eVONet, enet.py:218-233): `patchify` (fnet, inet, scorer — devo_amd.patchifier)
    and `update` (devo_amd.update.Update), same names."""

    def __init__(self, p=3, dim=384):
        super().__init__()
        from .update import Update
        from .patchifier import Patchifier
        self.patchify = Patchifier(patch_size=p, dim_inet=dim, dim_fnet=128, dim=32, patch_selector="scorer")
        self.update = Update(p, dim)
        self.P, self.dim = p, dim

    def num_parameters(self):
        return sum(q.numel() for q in self.parameters())

    def forward(self, batch, iters=18, corr_dropout=0.2, flow_weight=0.1, pose_weight=10.0):
        """One sequence (batch 1) through `iters` update iterations on its patch graph -> scalar loss
        (enet.py:300-370 on a fixed full graph + train.py:172-236)."""
        if "wiring_check" in batch:
            # not a training step: sum(parameters) * factor, so that a CPU / gloo run can check the DDP wiring of THIS module
            # (every parameter in the all-reduced bucket) without the HIP kernels — tests/test_distributed_gloo.py
            return sum((q * float(batch["wiring_check"])).sum() for q in self.parameters())
        from . import altcorr, projective_ops as pops
        from .ba import BA
        from .lietorch import SE3
        b = batch
        ii, jj, kk = b["ii"], b["jj"], b["kk"]
        E, n = ii.numel(), b["poses_gt"].shape[1]
        # enet.py:279-291: features, patch gathers and scores from the voxel grids; the patch centres are the synthetic
        # trajectory's (so that the ground-truth patches of the flow loss belong to them), the scorer is evaluated there
        fmap, gmap, imap, _, _, scores = self.patchify(b["images"], b["M"], coords=b["centres"])
        pyramid = [altcorr.channels_last(fmap), altcorr.channels_last(torch.nn.functional.avg_pool2d(fmap[0], 4, 4)[None])]   # enet.py:207-210
        imap = imap.view(1, -1, self.dim)
        Ps = SE3(b["poses_gt"])
        Gs = SE3(b["poses0"].clone())
        patches = b["patches0"].clone()
        net = torch.zeros(1, E, self.dim, device=ii.device)
        inp = torch.index_select(imap, 1, kk)
        bounds = [-64, -64, b["W"] + 64, b["H"] + 64]
        dij = (ii - jj).abs()
        close = (dij > 0) & (dij <= 2)
        ci, cj, ck = ii[close], jj[close], kk[close]
        with torch.no_grad():
            coords_gt, valid_gt = pops.transform(Ps, b["patches_gt"], b["intr"], ci, cj, ck, valid=True)[:2]
        fi, fj = torch.meshgrid(torch.arange(n, device=ii.device), torch.arange(n, device=ii.device), indexing="ij")
        fk = fi != fj
        fi, fj = fi[fk], fj[fk]
        loss = 0.0
        for it in range(iters):
            Gs = SE3(Gs.data.detach())
            patches = patches.detach()
            coords = pops.transform(Gs, patches, b["intr"], ii, jj, kk)
            coords1 = coords.permute(0, 1, 4, 2, 3).contiguous()
            if FUSED_LOOKUP:                                           # enet.py:203-216 as one launch (altcorr.CorrPyramidLayer)
                corr = altcorr.corr_pyramid(gmap, pyramid, coords1, kk, jj, b["R"], (1, 4), dropout=corr_dropout)
            else:
                corr = torch.stack([altcorr.corr(gmap, pyramid[0], coords1 / 1, kk, jj, b["R"], corr_dropout),
                                    altcorr.corr(gmap, pyramid[1], coords1 / 4, kk, jj, b["R"], corr_dropout)], -1).view(1, E, -1)
            net, (delta, weight, _) = self.update(net, inp, corr, None, ii, jj, kk)
            target = coords[..., self.P // 2, self.P // 2, :] + delta
            for _ in range(2):
                Gs, patches = BA(Gs, patches, b["intr"], target, weight, 1e-4, ii, jj, kk, bounds, ep=10.0, fixedp=1, n_frames=n)
            # flow loss over the close edges (train.py:177-181), pose loss over all frame pairs (:199-225, without the scale alignment)
            cf = pops.transform(Gs, patches, b["intr"], ci, cj, ck)
            e = (cf - coords_gt).norm(dim=-1).reshape(-1, self.P * self.P)
            ok = valid_gt.reshape(-1) > 0.5
            flow_loss = (e.min(dim=-1).values * ok).sum() / ok.sum().clamp(min=1)
            P1, P2 = Gs.inv(), Ps.inv()
            take = lambda G, idx: SE3(torch.index_select(G.data, 1, idx))
            dP = take(P1, fi).inv() * take(P1, fj)
            dG = take(P2, fi).inv() * take(P2, fj)
            e1 = (dP * dG.inv()).log()
            pose_loss = e1[..., 0:3].norm(dim=-1).mean() + e1[..., 3:6].norm(dim=-1).mean()
            loss = loss + flow_weight * flow_loss
            if it >= 2:
                loss = loss + pose_weight * pose_loss
        return loss + 1e-3 * scores.mean()                      # (the reference's scorer term, train.py:226-232, reduced to a mean)


def make_batch(workload="cfg2_m80", seed=1234, device="cuda"):
    """One synthetic training sequence: features, patches with perturbed depths, identity-initialised poses,
    ground-truth poses / patches, the full patch graph."""
    cfg = synth.workload(workload)
    n, M, H, W, C, R = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["C"], cfg["R"]
    dev = torch.device(device)
    from . import altcorr
    poses_gt = synth.make_poses(n, seed)
    patches_gt, centres = synth.make_patches(n, M, H, W, seed=seed)
    intr = synth.make_intrinsics(n, H, W)
    ii, jj, kk = synth.full_graph(n, M)
    g = torch.Generator().manual_seed(seed + 7)
    images = torch.randn(1, n, 5, 4 * H, 4 * W, generator=g)                                  # event voxel grids, 5 bins (std-normalised)
    patches0 = patches_gt.clone()
    patches0[:, :, 2] = torch.rand(1, n * M, 1, 1, generator=g).expand(1, n * M, 3, 3)       # enet.py:294-295: random initial depth
    poses0 = poses_gt.clone()
    poses0[:, 1:, :3] += 0.01 * torch.randn(1, n - 1, 3, generator=g)                         # start near, not at, the truth
    d = lambda t: t.to(dev)
    cx = patches_gt[0, :, 0, 1, 1].round().long().view(n, M).clamp(1, W - 2)                  # patch centres, feature-map pixels
    cy = patches_gt[0, :, 1, 1, 1].round().long().view(n, M).clamp(1, H - 2)
    return dict(images=d(images), centres=(d(cx), d(cy)),
                poses_gt=d(poses_gt), poses0=d(poses0), patches_gt=d(patches_gt), patches0=d(patches0), intr=d(intr),
                ii=d(ii), jj=d(jj), kk=d(kk), H=H, W=W, R=R, n=n, M=M, E=int(ii.numel()))


def build_trainer(device, world_size, lr=8e-5, seed=0, ddp=None):
    """net (DDP-wrapped when world_size > 1, train.py:106-107; ddp=True: also at world size 1, given a process group), AdamW (train.py:109)."""
    torch.manual_seed(seed)                                   # identical initial weights on every rank (train.py:41)
    net = TrainNet().to(device).train()
    model = net
    if world_size > 1 or ddp:
        from torch.nn.parallel import DistributedDataParallel as DDP
        dev = torch.device(device)
        model = DDP(net, device_ids=[dev.index] if dev.type == "cuda" else None, find_unused_parameters=False)
    opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=1e-6)
    return net, model, opt


def train_step(model, opt, batch, iters=18, clip=10.0):
    """optimizer.zero_grad -> forward -> backward (DDP: gradient all-reduce) -> clip -> step (train.py:166-250)."""
    opt.zero_grad(set_to_none=True)
    loss = model(batch, iters=iters)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
    opt.step()
    return loss.detach()
