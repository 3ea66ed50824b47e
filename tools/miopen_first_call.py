#!/usr/bin/env python3
"""What MIOpen's solver search costs the FIRST forward + backward of the Patchifier (15 voxel grids 480 x 640) on a fresh box, and the
steady-state step after it.  MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR pointing at empty directories show the cold cost (121 s);
the defaults set by devo_amd.patchifier use the DB shipped in devo_amd/miopen_db (0.5 s).   python tools/miopen_first_call.py"""
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
