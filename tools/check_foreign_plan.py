import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bench
from devo_amd import synth
from devo_amd.backends import cuda_ba, cuda_corr
dev = torch.device("cuda", 0)
for workload, dt in (("cfg2", torch.float32), ("cfg2", torch.float16), ("stress", torch.float32)):
    cfg = synth.workload(workload)
    d, cpu = bench.build_inputs(cfg, 1234, dev, dt, "blk8")
    coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
    n, R, H, W = cfg["n"], cfg["R"], cfg["H"], cfg["W"]
    E = d["ii"].numel()
    g = torch.Generator(device="cpu").manual_seed(0)
    other = coords + (torch.randn(coords.shape[0], E, 2, 1, 1, generator=g) * 40).to(dev)      # other boxes: other classes, other bins
    far = coords.clone(); far[:, ::3] += 5000.0                                                   # a third of the edges far outside the frame
    plan = cuda_corr.plan(coords, d["jj"], n, H, radius=R)
    look = lambda order: cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4), order=order)
    out = look(plan)
    for name, c2 in (("shifted", other), ("far", far)):
        p2 = cuda_corr.plan(c2, d["jj"], n, H, radius=R)
        o2 = look(p2)
        print(workload, dt, name, "heavy/dead own", int(plan[E]), int(plan[2 * E + 1]), "foreign", int(p2[E]), int(p2[2 * E + 1]), "identical", torch.equal(out, o2), flush=True)
    # per-level launches with a foreign single-level plan
    per = (2 * R + 1) ** 2 * 9
    for lvl, s in ((0, 1.0), (1, 4.0)):
        fm = d["pyramid"][lvl]
        o_own = torch.empty(1, E, per, dtype=dt, device=dev); o_for = torch.empty_like(o_own)
        cuda_corr.forward_into(o_own, d["gmap"], fm, coords, d["kk"], d["jj"], R, per, 1, 0, order=None, coord_div=s)
        pf = cuda_corr.plan(other, d["jj"], n, H // (4 if lvl else 1), float(s), R)
        p0 = cuda_corr.plan(coords, d["jj"], n, H, 1.0, R)                                         # the level-0 plan used for level 1 as well
        cuda_corr.forward_into(o_for, d["gmap"], fm, coords, d["kk"], d["jj"], R, per, 1, 0, order=pf, coord_div=s)
        o_p0 = torch.empty_like(o_own)
        cuda_corr.forward_into(o_p0, d["gmap"], fm, coords, d["kk"], d["jj"], R, per, 1, 0, order=p0, coord_div=s)
        print(workload, dt, "level", lvl, "foreign plan identical", torch.equal(o_own, o_for), "| level-0 plan identical", torch.equal(o_own, o_p0), flush=True)
