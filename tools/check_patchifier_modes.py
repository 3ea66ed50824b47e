import sys; sys.path.insert(0, "/root/repo")
import torch
from devo_amd.patchifier import Patchifier
pf = Patchifier().cuda().eval()
x = torch.randn(1, 1, 5, 96, 128, device="cuda")
outs = []
for mode in (torch.no_grad, torch.inference_mode):
    for _ in range(4):
        with mode(), torch.autocast("cuda", dtype=torch.float16):
            o = pf(x, 12, scorer_eval_mode="topk")
    outs.append(o)
print("inference_mode ok:", all(torch.equal(a, b) for a, b in zip(outs[0][:4], outs[1][:4])))
# a second stream
s = torch.cuda.Stream()
with torch.cuda.stream(s), torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    o2 = pf(x, 12, scorer_eval_mode="topk")
s.synchronize()
print("other stream ok:", all(torch.equal(a, b) for a, b in zip(outs[0][:4], o2[:4])))
