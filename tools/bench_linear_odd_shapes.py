import sys; sys.path.insert(0, "/root/repo")
import torch
from devo_amd import update as U
dev = torch.device("cuda", 0)
def timed(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rows, n_out, k_in in [(18000, 2, 384), (18000, 384, 2), (18000, 882, 384), (18000, 384, 882)]:
    x = torch.randn(rows, k_in, device=dev); w = torch.randn(n_out, k_in, device=dev) / k_in ** 0.5; b = torch.randn(n_out, device=dev)
    print(rows, n_out, k_in, "library %.1f us, split %.1f us" % (timed(lambda: torch.nn.functional.linear(x, w, b)), timed(lambda: U._linear_split(x, w, b))))
