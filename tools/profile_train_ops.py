#!/usr/bin/env python3
"""torch.profiler over one training step (3 update iterations): GPU time by ATen / autograd operator and input shape — which host-side
expressions the element-wise / copy kernels of tools/trace_categories.py belong to.  python tools/profile_train_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from devo_amd import training as T
dev = torch.device("cuda", 0)
net, model, opt = T.build_trainer(dev, 1)
batch = T.make_batch("cfg2_m80", 1234, dev)
ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(2):
    T.train_step(model, opt, batch, iters=ITERS)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    T.train_step(model, opt, batch, iters=ITERS)
    torch.cuda.synchronize()
ops = [e for e in prof.key_averages() if e.key.startswith("aten::") or "Backward" in e.key or e.key.startswith("_")]
ops.sort(key=lambda e: -e.self_device_time_total)
print("operators by their own GPU time (us), calls:")
for e in ops[:40]:
    print(f"  {e.key[:44]:44s} {e.self_device_time_total:10.0f} {e.count:6d}")
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=48, max_shapes_column_width=70))
print("the element-wise / reduction operators by input shape:")
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key in ("aten::sum", "aten::add_", "aten::add", "aten::copy_", "aten::threshold_backward", "aten::gather", "aten::index_add_", "aten::mul", "aten::div", "aten::fill_", "aten::clamp_min", "aten::mm", "aten::bmm", "aten::addmm")]
rows.sort(key=lambda e: -e.self_device_time_total)
for e in rows[:45]:
    print(f"  {e.key:26s} {e.self_device_time_total:9.0f} us {e.count:5d} calls  {str(e.input_shapes)[:120]}")
