#!/usr/bin/env python3
"""Per-frame cost of the Patchifier on the GPU: encoders (MIOpen, channels-last) + scorer + selection + HIP
gathers + both pyramid levels, at DEVO's input size (5-bin voxel grid 480 x 640 -> 120 x 160 features, 96 patches).
python tools/bench_patchifier.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd.patchifier import Patchifier

dev = torch.device("cuda", 0)
torch.manual_seed(0)
pf = Patchifier().to(dev).eval()
H, W, M = 480, 640, 96


def timed(fn, reps=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for n in (1, 15):
    images = torch.randn(1, n, 5, H, W, device=dev)
    for tag, ctx in (("fp32", lambda: torch.autocast("cuda", enabled=False)), ("autocast fp16", lambda: torch.autocast("cuda", dtype=torch.float16)),
                     ("autocast bf16", lambda: torch.autocast("cuda", dtype=torch.bfloat16))):
        def run():
            with torch.no_grad(), ctx():
                fmap, gmap, imap, patches, index = pf(images, M, scorer_eval_mode="topk")
            return pf.pyramid(fmap.float() if tag == "fp32" else fmap.half())
        ms = timed(run)
        def enc():
            with torch.no_grad(), ctx():
                return pf.fnet(images), pf.inet(images), pf.scorer(images)
        ms_enc = timed(enc)
        print(f"{n:2d} frame(s), {tag:14s}: patchify + pyramid {ms:7.3f} ms  (the three CNNs alone {ms_enc:7.3f} ms)", flush=True)
pf.train()
images = torch.randn(1, 15, 5, H, W, device=dev)
def step():
    for q in pf.parameters(): q.grad = None
    fmap, gmap, imap, patches, index, scores = pf(images, 80)
    (fmap.square().mean() + gmap.square().mean() + imap.square().mean() + scores.mean()).backward()
print(f"15 frames, fp32, training forward + backward: {timed(step, reps=5, warm=2):7.3f} ms")
