import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from devo_amd.ba import BA
from devo_amd import projective_ops as pops
from devo_amd.lietorch import SE3
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'ba_train_f64.npz'))
def load(dt):
    g = {}
    for k in z.files:
        if z[k].ndim == 0: g[k] = z[k].item()
        else:
            t = torch.from_numpy(z[k]); g[k] = (t.to(dt) if t.is_floating_point() else t).to('cuda')
    return g
def run(dt, torch_path):
    os.environ["DEVO_BA_TORCH"] = "1" if torch_path else "0"
    g = load(dt)
    tgt = g["target"].clone().requires_grad_(True); wgt = g["weight"].clone().requires_grad_(True)
    G, P = BA(SE3(g["poses"].clone()), g["patches"].clone(), g["intrinsics"], tgt, wgt, 1e-4, g["ii"], g["jj"], g["kk"], g["bounds"].tolist(), ep=10.0, fixedp=1)
    cf = pops.transform(G, P, g["intrinsics"], g["ii"], g["jj"], g["kk"])
    loss = (cf * g["loss_weights"]).sum() + (G.log() ** 2).sum()
    loss.backward()
    return tgt.grad.double().cpu(), wgt.grad.double().cpu(), g
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
t64, w64, g = run(torch.float64, True)
gt, gw = g["grad_target"].double().cpu(), g["grad_weight"].double().cpu()
print("fp64 torch vs golden:", rel(t64, gt), rel(w64, gw))
for tp in (True, False):
    t32, w32, _ = run(torch.float32, tp)
    print("fp32", "torch" if tp else "fused", "vs golden: target", rel(t32, gt), "weight", rel(w32, gw))
