#!/usr/bin/env python3
"""One launch of the gemm_rs kernel per shape (DEVO_RS_TRACE=1 prints its phases; rocprofv3 --pmc counts it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from devo_amd import update as UA
dev = torch.device("cuda", 0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 21600
torch.manual_seed(0)
for N in (384, 768):
    lin = torch.nn.Linear(384, N).to(dev).half()
    x = (torch.randn(rows, 384, device=dev) * 0.5).half()
    for _ in range(3):
        UA._linear_f16(x, lin.weight, lin.bias)
    torch.cuda.synchronize()
