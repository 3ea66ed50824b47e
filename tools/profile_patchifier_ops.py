import os, sys
sys.path.insert(0, "/root/repo")
import torch
from devo_amd.patchifier import Patchifier
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
torch.manual_seed(0)
pf = Patchifier().to(dev).eval()
images = torch.randn(1, 1, 5, 480, 640, device=dev)
def run():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        return pf(images, 96)
os.environ["X"]="1"
import devo_amd.patchifier as PFm
PFm._GRAPH = False          # eager launches: the profiler then attributes kernels to ops
for _ in range(5): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(5): run()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="device_time_total", row_limit=22, max_name_column_width=40, max_shapes_column_width=60)[:9000])
