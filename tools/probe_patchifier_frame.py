import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from devo_amd.patchifier import Patchifier
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pf = Patchifier().to(dev).eval()
images = torch.randn(1, 1, 5, 480, 640, device=dev)
def run():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        return pf(images, 96, scorer_eval_mode="topk")
for _ in range(5): run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): run()
th = time.perf_counter() - t0
torch.cuda.synchronize()
ta = time.perf_counter() - t0
print(f"eager: host {th/20*1e3:.3f} ms, wall {ta/20*1e3:.3f} ms")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5): run()
    torch.cuda.synchronize()
tot = 0; n = 0; names = {}
for ev in prof.events():
    if "cuda" in str(getattr(ev, "device_type", "")).lower():
        d = float(getattr(ev, "device_time", 0.0) or 0.0); tot += d; n += 1
        names[ev.name[:70]] = names.get(ev.name[:70], 0) + d
print(f"kernels per call {n/5:.0f}, GPU busy per call {tot/5/1e3:.3f} ms")
for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:14]:
    print(f"  {v/5:8.1f} us  {k}")
# graph capture
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
        with torch.cuda.graph(g, stream=s):
            out = run()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {(time.perf_counter()-t0)/20*1e3:.3f} ms")
except Exception as ex:
    print("graph capture failed:", type(ex).__name__, str(ex)[:300])
