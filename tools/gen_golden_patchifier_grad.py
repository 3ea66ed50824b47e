#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator; build container only — it imports the reference and oracle/).

tests/golden/patchifier_grad_f64.npz: the REAL reference devo.enet.Patchifier (training mode, scorer selection; reduced width, CPU, fp64)
with a loss on everything it returns — fmap, the gathered gmap / imap patches, the winners' scores — and the gradients of that loss with
respect to parameters of both encoders and of the scorer.  Pins the training path of SURVEY 8f row f3 (the encoders' and the scorer's
autograd through the patch gathers and the score lookup), which so far was compared on values only.
Shims: tools/gen_golden_patchifier.py's (device arguments dropped, the random candidates recorded), plus
cuda_corr.patchify_backward = oracle/altcorr.py:patchify_backward (restatement of correlation_kernel.cu:49-80)."""
import os
import sys
import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_golden_update as G                          # noqa: E402
from oracle import altcorr as A                        # noqa: E402

DRAWS = []


def main():
    G.install_shims()
    sys.modules["cuda_corr"].patchify_forward = lambda net, coords, radius: [A.patchify_forward(net, coords, radius)]
    sys.modules["cuda_corr"].patchify_backward = lambda net, coords, grad, radius: [A.patchify_backward(net, coords, grad, radius)]
    for name in ("randint", "ones", "arange", "as_tensor"):
        orig = getattr(torch, name)

        def wrapped(*a, _orig=orig, _name=name, **k):
            k.pop("device", None)
            r = _orig(*a, **k)
            if _name == "randint":
                DRAWS.append(r.clone())
            return r
        setattr(torch, name, wrapped)
    from devo.enet import Patchifier
    torch.manual_seed(8765)
    g = torch.Generator().manual_seed(21)
    images = (torch.randn(1, 2, 5, 48, 64, generator=g) * 2.0).float().double()
    pf = Patchifier(patch_size=3, dim_inet=24, dim_fnet=16, dim=8, patch_selector="scorer").double().train()
    with torch.no_grad():                                  # (the freshly initialised scorer saturates its sigmoid on these inputs: every score 1.0, the
        pf.scorer.scorer[6].weight.mul_(0.02)              #  winners a matter of tie-breaking — a smaller last layer spreads the scores)
        pf.scorer.scorer[6].bias.zero_()
    out = {"images": images.numpy().astype(np.float32)}
    for k, v in pf.state_dict().items():
        out["sd/" + k] = v.numpy().astype(np.float32)
    pf.load_state_dict({k: v.float().double() for k, v in pf.state_dict().items()})          # fp32-representable weights, fp64 arithmetic
    DRAWS.clear()
    fmap, gmap, imap, patches, index, scores = pf(images, patches_per_image=6)
    out["cand_x"], out["cand_y"] = DRAWS[0].numpy(), DRAWS[1].numpy()
    w = [torch.randn(t.shape, generator=g, dtype=torch.float64) for t in (fmap, gmap, imap, scores)]
    loss = (fmap * w[0]).sum() * 1e-2 + (gmap * w[1]).sum() + (imap * w[2]).sum() + (scores * w[3]).sum()
    loss.backward()
    for k, t in zip(("fmap", "gmap", "imap", "scores"), w):
        out["w/" + k] = t.numpy().astype(np.float32)
    out["loss"] = float(loss.detach())
    for k, t in dict(fmap=fmap, gmap=gmap, imap=imap, patches=patches, scores=scores).items():
        out["out/" + k] = t.detach().numpy().astype(np.float32)
    grads = {k: v.grad.numpy() for k, v in pf.named_parameters() if v.grad is not None}
    out["grad_names"] = np.array(sorted(grads))
    out["grad_norms"] = np.array([float(np.linalg.norm(grads[k])) for k in sorted(grads)])
    keep = [k for k in ("fnet.conv1.weight", "inet.conv1.weight", "fnet.conv2.weight", "inet.conv2.weight", "scorer.scorer.0.weight", "scorer.scorer.6.weight") if k in grads]
    for k in keep:
        out["grad/" + k] = grads[k].astype(np.float64)
    path = os.path.join(ROOT, "tests", "golden", "patchifier_grad_f64.npz")
    np.savez_compressed(path, **out)
    sc = scores.detach()
    assert float(sc.max() - sc.min()) > 1e-3 and len(set(sc.flatten().tolist())) == sc.numel(), "degenerate scores"
    print(path, os.path.getsize(path) // 1024, "KB; scores", float(sc.min()), float(sc.max()), "loss", out["loss"], "params with grad", len(grads), "stored", keep)


if __name__ == "__main__":
    main()
