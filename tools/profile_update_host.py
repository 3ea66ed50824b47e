"""Host side of the fp16 Update operator: enqueue time per call against the wall time, and a cProfile of 200 calls (the operator is
host-bound in eager mode when its Python side costs more than the GPU's 0.23 ms)."""
import sys, time, cProfile, pstats, io, torch
sys.path.insert(0, "/root/repo")
from devo_amd import synth
from devo_amd.update import Update
dev = torch.device("cuda", 0)
ii, jj, kk = (t.to(dev) for t in synth.full_graph(15, 96)); E = ii.numel()
torch.manual_seed(0)
m = Update(3).to(dev).half().eval()
net, inp = torch.randn(1, E, 384, device=dev).half(), torch.randn(1, E, 384, device=dev).half(); corr = torch.randn(1, E, 882, device=dev).half()
with torch.no_grad():
    for _ in range(5): m(net, inp, corr, None, ii, jj, kk)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): m(net, inp, corr, None, ii, jj, kk)
    th = time.perf_counter() - t0; torch.cuda.synchronize(); ta = time.perf_counter() - t0
    print(f"host {th * 1e4:.1f} us per call, wall {ta * 1e4:.1f}")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): m(net, inp, corr, None, ii, jj, kk)
    pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
