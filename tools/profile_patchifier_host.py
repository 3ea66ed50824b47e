import os, sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch
from devo_amd.patchifier import Patchifier
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pf = Patchifier().to(dev).eval()
images = torch.randn(1, 1, 5, 480, 640, device=dev)
def run():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        return pf(images, 96, scorer_eval_mode=os.environ.get("MODE", "multi"))
for _ in range(6): run()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(30): run()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print("\n".join(l[:150] for l in s.getvalue().splitlines()[:40]))
