import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from devo_amd.patchifier import Patchifier
dev = torch.device("cuda", 0)
pf = Patchifier().to(dev).train()
images = torch.randn(1, 15, 5, 480, 640, device=dev)
def step():
    for q in pf.parameters(): q.grad = None
    out = pf(images, 80)
    (out[0].square().mean() + out[1].square().mean() + out[2].square().mean() + out[5].mean()).backward()
    torch.cuda.synchronize()
t0 = time.perf_counter(); step(); t1 = time.perf_counter()
for _ in range(2): step()
t2 = time.perf_counter()
for _ in range(5): step()
t3 = time.perf_counter()
print(f"MIOPEN_FIND_MODE={os.environ.get('MIOPEN_FIND_MODE')}: first step {t1 - t0:.2f} s, steady {1e3 * (t3 - t2) / 5:.1f} ms")
