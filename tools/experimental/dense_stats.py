"""Round statistics of the region-staged lookup kernel at a bench workload (DEVO_DN_STATS=1 python tools/dense_stats.py [cfg2|stress] [f16|f32])."""
import os, sys
os.environ["DEVO_DN_STATS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import synth
from devo_amd.backends import cuda_ba, cuda_corr
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else torch.float32
cfg = synth.workload(wl)
dev = torch.device("cuda", 0)
d, _ = bench.build_inputs(cfg, 1234, dev, dt, "blk8")
coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
for use_plan in (True, False):
    order = cuda_corr.plan(coords, d["jj"], cfg["n"], cfg["H"], 1.0, cfg["R"]) if use_plan else torch.arange(2 * coords.shape[1] + 1, dtype=torch.int32, device=dev)
    print("with plan" if use_plan else "identity order", flush=True)
    cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], cfg["R"], (1, 4), order=order)
    torch.cuda.synchronize()
print("per-level launches (NL = 1 instantiation)", flush=True)
order = cuda_corr.plan(coords, d["jj"], cfg["n"], cfg["H"], 1.0, cfg["R"])
Dm = 2 * cfg["R"] + 1
out = torch.empty(1, coords.shape[1], Dm * Dm * 18, dtype=dt, device=dev)
for lvl, s in enumerate((1.0, 4.0)):
    cuda_corr.forward_into(out, d["gmap"], d["pyramid"][lvl], coords, d["kk"], d["jj"], cfg["R"], Dm * Dm * 18, 2, lvl, order=order, coord_div=s)
    torch.cuda.synchronize()
