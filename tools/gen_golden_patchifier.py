#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (fixture generator; build container only — it imports the reference and oracle/).

Generate tests/golden/patchifier_f64.npz by running the REAL reference modules from /root/reference on CPU in fp64 at a reduced
width (dim = 8, output dims 16 / 24) so that the fixture stays small:
  devo.extractor.BasicEncoder4Evs (norm_fn 'instance' and 'none'), devo.selector.Scorer, devo.selector.PatchSelector("topk"),
  and devo.enet.Patchifier.forward in training mode with the scorer (3 M random candidates, top-M by score).
Nothing from the reference is copied: the file holds data (seeded weights from the modules' own initialisers, inputs, the random
candidates the reference drew, expected outputs).

Shims installed before the imports (gen_golden_update.py's set, plus):
  cuda_corr.patchify_forward  = oracle/altcorr.py:patchify_forward (restatement of correlation_kernel.cu:16-47)
  torch.randint / torch.ones / torch.arange / torch.as_tensor: `device="cuda"` arguments are dropped (no GPU in this container;
  the reference hard-codes the device, enet.py:139-147,184-190), and every randint draw is recorded.
"""
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
    sys.modules["torch_scatter"].__dict__.setdefault("__version__", "0")
    for name in ("randint", "ones", "arange", "as_tensor"):
        orig = getattr(torch, name)

        def wrapped(*a, _orig=orig, _name=name, **k):
            k.pop("device", None)
            r = _orig(*a, **k)
            if _name == "randint":
                DRAWS.append(r.clone())
            return r
        setattr(torch, name, wrapped)
    from devo.extractor import BasicEncoder4Evs
    from devo.selector import Scorer, PatchSelector
    from devo.enet import Patchifier

    torch.manual_seed(4321)
    g = torch.Generator().manual_seed(11)
    out = {}
    images = torch.randn(1, 2, 5, 48, 64, generator=g).double() * 2.0
    out["images"] = images.float().numpy()
    images = images.float().double()                    # fp32-representable inputs, fp64 arithmetic
    for tag, norm, od in (("fnet", "instance", 16), ("inet", "none", 24)):
        enc = BasicEncoder4Evs(output_dim=od, dim=8, norm_fn=norm).double().eval()
        for k, v in enc.state_dict().items():
            out[f"{tag}/sd/{k}"] = v.numpy()
        with torch.no_grad():
            out[f"{tag}/out"] = enc(images).numpy()
    sc = Scorer(5).double().eval()
    for k, v in sc.state_dict().items():
        out["scorer/sd/" + k] = v.numpy()
    with torch.no_grad():
        smap = sc(images)
    out["scorer/out"] = smap.numpy()
    # pooled top-k on a score map (selector.py:152-192), with and without the 2 x 2 grid
    sm = torch.rand(1, 3, 22, 30, generator=g).double()
    out["topk/scores"] = sm.numpy()
    for grid in (True, False):
        x, y = PatchSelector("topk", grid=grid)(sm, 8)
        out[f"topk/x_grid{int(grid)}"] = x.numpy(); out[f"topk/y_grid{int(grid)}"] = y.numpy()

    # the whole module, training mode, scorer selection (enet.py:120-200)
    pf = Patchifier(patch_size=3, dim_inet=24, dim_fnet=16, dim=8, patch_selector="scorer").double().train()
    for k, v in pf.state_dict().items():
        out["pf/sd/" + k] = v.numpy()
    DRAWS.clear()
    with torch.no_grad():
        fmap, gmap, imap, patches, index, scores = pf(images, patches_per_image=6)
    out["pf/cand_x"], out["pf/cand_y"] = DRAWS[0].numpy(), DRAWS[1].numpy()
    for k, v in dict(fmap=fmap, gmap=gmap, imap=imap, patches=patches, index=index, scores=scores).items():
        out["pf/" + k] = v.numpy()
    # stored as fp32 (weights and inputs are fp32-representable; expected outputs round at 6e-8, far below the tests' tolerance)
    out = {k: (v.astype(np.float32) if v.dtype == np.float64 else v) for k, v in out.items()}
    path = os.path.join(ROOT, "tests", "golden", "patchifier_f64.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
