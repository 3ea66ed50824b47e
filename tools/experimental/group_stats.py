"""Phase statistics of the edge-group lookup kernel at a bench workload (python tools/group_stats.py [cfg2|stress] [reps])."""
import os, sys
os.environ["DEVO_GP_STATS"] = "1"
os.environ["DEVO_CORR_GROUP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import synth
from devo_amd.backends import cuda_ba, cuda_corr
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = synth.workload(wl)
dev = torch.device("cuda", 0)
d, _ = bench.build_inputs(cfg, 1234, dev, torch.float16, "blk8")
coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
for use_plan in (True, False):
    order = cuda_corr.plan(coords, d["jj"], cfg["n"], cfg["H"], 1.0, cfg["R"]) if use_plan else torch.arange(2 * coords.shape[1] + 1, dtype=torch.int32, device=dev)
    print("with plan" if use_plan else "identity order", flush=True)
    for _ in range(2):
        cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], cfg["R"], (1, 4), order=order)
        torch.cuda.synchronize()
