#!/usr/bin/env python3
"""Where the launches of a training iteration come from: one forward + backward of devo_amd.training.TrainNet with 3 update iterations,
every section of the iteration body under a record_function label; per label the number of GPU kernels and their time, forward and (via the
autograd sequence numbers the backward nodes inherit) backward."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity, record_function
from devo_amd import training as T, altcorr, projective_ops as pops
from devo_amd.ba import BA
from devo_amd.lietorch import SE3

dev = "cuda"
net, model, opt = T.build_trainer(dev, 1)
b = T.make_batch("cfg2_m80", 1234, dev)


def body(iters=3):
    self = net
    ii, jj, kk = b["ii"], b["jj"], b["kk"]
    E, n = ii.numel(), b["poses_gt"].shape[1]
    with record_function("S:patchify"):
        fmap, gmap, imap, _, _, scores = self.patchify(b["images"], b["M"], coords=b["centres"])
        pyramid = [altcorr.channels_last(fmap), altcorr.channels_last(torch.nn.functional.avg_pool2d(fmap[0], 4, 4)[None])]
        imap = imap.view(1, -1, self.dim)
    Ps = SE3(b["poses_gt"]); Gs = SE3(b["poses0"].clone()); patches = b["patches0"].clone()
    netst = torch.zeros(1, E, self.dim, device=ii.device)
    inp = torch.index_select(imap, 1, kk)
    bounds = [-64, -64, b["W"] + 64, b["H"] + 64]
    dij = (ii - jj).abs(); close = (dij > 0) & (dij <= 2)
    ci, cj, ck = ii[close], jj[close], kk[close]
    with torch.no_grad():
        coords_gt, valid_gt = pops.transform(Ps, b["patches_gt"], b["intr"], ci, cj, ck, valid=True)[:2]
    fi, fj = torch.meshgrid(torch.arange(n, device=ii.device), torch.arange(n, device=ii.device), indexing="ij")
    fk = fi != fj; fi, fj = fi[fk], fj[fk]
    loss = 0.0
    for it in range(iters):
        with record_function("S:detach+transform"):
            Gs = SE3(Gs.data.detach()); patches = patches.detach()
            coords = pops.transform(Gs, patches, b["intr"], ii, jj, kk)
            coords1 = coords.permute(0, 1, 4, 2, 3).contiguous()
        with record_function("S:lookup"):
            corr = altcorr.corr_pyramid(gmap, pyramid, coords1, kk, jj, b["R"], (1, 4), dropout=0.2)
        with record_function("S:update"):
            netst, (delta, weight, _) = self.update(netst, inp, corr, None, ii, jj, kk)
        with record_function("S:target+BA"):
            target = coords[..., self.P // 2, self.P // 2, :] + delta
            for _ in range(2):
                Gs, patches = BA(Gs, patches, b["intr"], target, weight, 1e-4, ii, jj, kk, bounds, ep=10.0, fixedp=1, n_frames=n)
        with record_function("S:flow_loss"):
            cf = pops.transform(Gs, patches, b["intr"], ci, cj, ck)
            e = (cf - coords_gt).norm(dim=-1).reshape(-1, self.P * self.P)
            ok = valid_gt.reshape(-1) > 0.5
            flow_loss = (e.min(dim=-1).values * ok).sum() / ok.sum().clamp(min=1)
        with record_function("S:pose_loss"):
            P1, P2 = Gs.inv(), Ps.inv()
            take = lambda G, idx: SE3(torch.index_select(G.data, 1, idx))
            dP = take(P1, fi).inv() * take(P1, fj)
            dG = take(P2, fi).inv() * take(P2, fj)
            e1 = (dP * dG.inv()).log()
            pose_loss = e1[..., 0:3].norm(dim=-1).mean() + e1[..., 3:6].norm(dim=-1).mean()
        with record_function("S:sum"):
            loss = loss + 0.1 * flow_loss
            if it >= 2:
                loss = loss + 10.0 * pose_loss
    return loss + 1e-3 * scores.mean()


for _ in range(2):
    opt.zero_grad(set_to_none=True); l = body(); l.backward()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    opt.zero_grad(set_to_none=True)
    with record_function("S:FORWARD"):
        l = body()
    with record_function("S:BACKWARD"):
        l.backward()
    torch.cuda.synchronize()
evs = prof.events()
# forward: kernels under each label; sequence numbers seen under each label -> backward kernels by sequence number
def label_of(e):
    p = e
    while p is not None:
        if p.name.startswith("S:") and p.name not in ("S:FORWARD", "S:BACKWARD"):
            return p.name
        p = p.cpu_parent
    return None
def top_of(e):
    p = e
    while p is not None:
        if p.name in ("S:FORWARD", "S:BACKWARD"):
            return p.name
        p = p.cpu_parent
    return None
fw = collections.defaultdict(lambda: [0, 0.0]); seq2label = {}
for e in evs:
    if str(e.device_type).endswith("CPU") and e.kernels:
        lab, top = label_of(e), top_of(e)
        if top == "S:FORWARD" and lab:
            fw[lab][0] += len(e.kernels); fw[lab][1] += sum(k.duration for k in e.kernels)
    if str(e.device_type).endswith("CPU") and e.sequence_nr is not None and e.sequence_nr >= 0 and top_of(e) == "S:FORWARD":
        lab = label_of(e)
        if lab: seq2label.setdefault(e.sequence_nr, lab)
bw = collections.defaultdict(lambda: [0, 0.0]); un = [0, 0.0]
names = collections.defaultdict(collections.Counter)
for e in evs:
    if str(e.device_type).endswith("CPU") and e.kernels and top_of(e) != "S:FORWARD":        # (the autograd engine's thread: no parent on this one)
        p, lab = e, None
        while p is not None and lab is None:
            if p.sequence_nr is not None and p.sequence_nr in seq2label and ("Backward" in p.name or "backward" in p.name):
                lab = seq2label[p.sequence_nr]
            p = p.cpu_parent
        tgt = bw[lab] if lab else un
        tgt[0] += len(e.kernels); tgt[1] += sum(k.duration for k in e.kernels)
        if lab in ("S:pose_loss", "S:flow_loss", "S:target+BA", "S:update"):
            q = e
            while q.cpu_parent is not None and "evaluate_function" not in q.name: q = q.cpu_parent
            names[lab][q.name.split(": ")[-1]] += len(e.kernels)
print("3 update iterations, forward + backward (kernels, GPU us):")
for lab in sorted(set(fw) | set(bw)):
    print(f"  {lab:22s} forward {fw[lab][0]:5d} kernels {fw[lab][1]:9.0f} us | backward {bw[lab][0]:5d} kernels {bw[lab][1]:9.0f} us")
print(f"  backward kernels not attributed: {un[0]} ({un[1]:.0f} us)")
for lab, c in names.items():
    print(f"  {lab} backward nodes by kernels:", ", ".join(f"{k} {v}" for k, v in c.most_common(14)))
