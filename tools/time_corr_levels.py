#!/usr/bin/env python3
"""Time the level-0 and level-1 lookups separately (median of 30 single launches, HIP events).
    LAYOUT=cl|blk8   DEVO_LIBS=a.so,b.so (A/B different builds of libdevo_hip.so in ONE gpurun call: box-to-box
    variation is ~1 %, larger than most kernel tweaks)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    import devo_amd._lib as L
    if os.environ.get("DEVO_LIB"): L.LIB_PATH = os.environ["DEVO_LIB"]
    from bench import build_inputs
    from devo_amd import synth
    from devo_amd.backends import cuda_ba, cuda_corr
    dev = torch.device("cuda", 0)
    cfg = synth.workload(os.environ.get("WL", "cfg2"))
    d, _ = build_inputs(cfg, 1234, dev, torch.float32, "cl")
    n, R = cfg["n"], cfg["R"]; E = d["ii"].numel(); Dm = 2 * R + 1
    out = torch.empty(1, E, Dm * Dm * 18, device=dev)
    coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
    order = cuda_corr.plan(coords, d["jj"], n, cfg["H"])
    from devo_amd import altcorr
    lay = os.environ.get("LAYOUT", "cl")
    ref = []
    for lvl in (0, 1):
        o = torch.zeros_like(out); cuda_corr.forward_into(o, d["gmap"], d["pyramid"][lvl], coords, d["kk"], d["jj"], R, Dm * Dm * 18, 2, lvl, order=order, coord_div=(1.0, 4.0)[lvl]); ref.append(o)
    if lay.startswith("blk"):
        d["pyramid"] = [altcorr.channel_blocked(f, int(lay[3:])) for f in d["pyramid"]]
        for lvl in (0, 1):
            o = torch.zeros_like(out); cuda_corr.forward_into(o, d["gmap"], d["pyramid"][lvl], coords, d["kk"], d["jj"], R, Dm * Dm * 18, 2, lvl, order=order, coord_div=(1.0, 4.0)[lvl])
            print("blocked vs cl equal:", torch.equal(o, ref[lvl]), float((o - ref[lvl]).abs().max()))
    res = []
    for lvl in (0, 1):
        fm, c_, dv = d["pyramid"][lvl], coords, (1.0, 4.0)[lvl]
        for _ in range(5): cuda_corr.forward_into(out, d["gmap"], fm, c_, d["kk"], d["jj"], R, Dm * Dm * 18, 2, lvl, order=order, coord_div=dv)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); cuda_corr.forward_into(out, d["gmap"], fm, c_, d["kk"], d["jj"], R, Dm * Dm * 18, 2, lvl, order=order, coord_div=dv); b.record()
            torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
        ts.sort(); res.append(ts[len(ts) // 2])
    print(f"{os.path.basename(os.environ.get('DEVO_LIB','default')):24s} {lay} nodma={os.environ.get('DEVO_CORR_NODMA','0')} L0 {res[0]:7.1f} us   L1 {res[1]:7.1f} us", flush=True)
else:
    for lib in os.environ.get("DEVO_LIBS", "").split(","):
        env = dict(os.environ); 
        if lib: env["DEVO_LIB"] = os.path.abspath(lib)
        subprocess.run([sys.executable, __file__, "--child"], env=env)
