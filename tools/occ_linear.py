import os, sys
sys.path.insert(0, "/root/repo")
import torch
from devo_amd import update as U
dev = torch.device("cuda", 0)
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
w = torch.randn(384, 384, device=dev) / 384 ** 0.5; b = torch.randn(384, device=dev)
for wgs in (64, 128, 256, 384, 512, 640, 768, 1024, 1536):
    rows = wgs // 4 * 128
    x = torch.randn(rows, 384, device=dev)
    print(wgs, "WGs", rows, "rows:", round(timed(lambda: U._linear_split(x, w, b)), 1), "us")
