import os, sys
os.environ["DEVO_LN_DBG"] = "48"
sys.path.insert(0, "/root/repo")
import torch
from devo_amd import update as U
dev = torch.device("cuda", 0)
for rows in (8192, 16384, 18000, 24576):
    x = torch.randn(rows, 384, device=dev); w = torch.randn(384, 384, device=dev) / 384 ** 0.5; b = torch.randn(384, device=dev)
    print("rows", rows, flush=True)
    for _ in range(2): y = U._linear_split(x, w, b)
    torch.cuda.synchronize()
