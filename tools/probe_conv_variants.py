#!/usr/bin/env python3
"""What one 3 x 3 convolution of the encoders costs through PyTorch -> MIOpen on a channels-last fp16 activation, by weight layout and bias:
kernels launched per call and their time (the igemm kernel is ~19 us of a 34 us call in the Patchifier's per-frame path)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
import devo_amd.patchifier                                   # (MIOpen's find db of this repo)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for (cin, cout, H, W, k, stride) in ((32, 32, 240, 320, 3, 1), (64, 64, 120, 160, 3, 1), (64, 128, 120, 160, 1, 1), (5, 32, 480, 640, 7, 2)):
    x = torch.randn(1, cin, H, W, device=dev).half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) * 0.05).half()
    b = torch.randn(cout, device=dev).half()
    for name, ww, bb in (("NCHW weight + bias", w, b), ("CL weight + bias", w.contiguous(memory_format=torch.channels_last), b),
                         ("NCHW weight, no bias", w, None), ("CL weight, no bias", w.contiguous(memory_format=torch.channels_last), None)):
        fn = lambda: F.conv2d(x, ww, bb, stride=stride, padding=k // 2)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(10): fn()
            torch.cuda.synchronize()
        ker = {}
        for ev in prof.events():
            if "cuda" in str(getattr(ev, "device_type", "")).lower():
                a = ker.setdefault(ev.name[:44], [0, 0.0]); a[0] += 1; a[1] += float(getattr(ev, "device_time", 0.0) or 0.0)
        tot = sum(v[1] for v in ker.values()) / 10
        print(f"{cin:3d}->{cout:3d} {H}x{W} k{k} s{stride}  {name:22s} {tot:7.1f} us/call: " + "; ".join(f"{n} x{v[0] // 10} {v[1] / 10:.1f}" for n, v in sorted(ker.items(), key=lambda kv: -kv[1][1])))
