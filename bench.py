#!/usr/bin/env python3
"""Benchmark of the DEVO update-op hot path on MI355X (BASELINE.json metric:
"update-op iterations/sec (altcorr+fastba) at 96 patches, N=15 keyframes").

One STEP = one update-op iteration on a fixed patch graph (devo/devo.py:308-338 minus the Update MLP):
    reproject (projective_ops.transform)  ->  altcorr lookup at 2 pyramid levels (r=3)
    ->  target = centre + delta (formed inside the BA)  ->  fastba bundle adjustment, 2 Gauss-Newton iterations.
All inputs are synthetic, resident in HBM before the timed region; the steps are captured in a
HIP graph (--steps-per-graph consecutive steps per graph launch, every step complete) and replayed.  Multi-GPU: one process per GPU, independent sequences (seed 1234 + rank), no data-path
collective (replicas; "scaling": "weak"); the aggregate is steps*ranks / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2] [--dtype f32|f16]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--layout", default="blk8", choices=["blk8", "cl", "nchw"],
                    help="storage of the feature pyramid: channel-blocked [n,C/8,H,W,8] / channels-last / the reference's NCHW")
    ap.add_argument("--per-level-launches", action="store_true",
                    help="one lookup launch per pyramid level instead of both levels in ONE launch (devo_corr_forward_pyramid2, what "
                         "altcorr.corr_pyramid does; measured 1.5 %% faster per step: same kernel time, one launch gap less)")
    ap.add_argument("--overlap-prepare", action="store_true",
                    help="run the BA index preparation (cuda_ba.prepare) on a side stream under the lookup instead of inside "
                         "cuda_ba.forward (measured slower on MI355X: the fork/join costs more than the 26 us it hides)")
    ap.add_argument("--separate-index-kernels", action="store_true",
                    help="the lookup plan's ordering kernel and the BA's index preparation as two launches (inside their own calls) "
                         "instead of one launch with two workgroups (cuda_ba.prepare(..., plan=...))")
    ap.add_argument("--plan", default="edges", choices=["groups", "edges"],
                    help="locality plan of the fused lookup: edges = the per-edge lookup (default); groups = edges sorted by (frame, tile of 6 x 6 "
                         "level-1 cells): level 1 is read from LDS regions shared by a group's edges (csrc/corr_mm.h group form; measured slower at "
                         "cfg2, profiles/r05_group_form.txt)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a HIP graph")
    ap.add_argument("--separate-target", action="store_true", help="form target = coords centre + delta with a torch kernel (devo.py:330) instead of inside the BA")
    ap.add_argument("--steps-per-graph", type=int, default=18,
                    help="consecutive steps captured into one HIP graph launch (a graph launch costs ~10 us of idle GPU; 1 = one launch per step).  Default 18 = "
                         "the update iterations DEVO_base.conf runs on one patch graph (BASELINE configuration 2)")
    ap.add_argument("--prepare-every", type=int, default=18,
                    help="the BA's index tables (unique patches, edges grouped by patch: a function of kk alone) are rebuilt every this many steps — 18 = once per "
                         "patch graph of 18 update iterations, what cuda_ba.forward's prepared-table cache does for an unchanged kk; 1 = every step (a new graph per step)")
    ap.add_argument("--plan-lag", type=int, default=1, choices=[0, 1],
                    help="1 (default): on an unchanged patch graph the lookup of update iteration k + 1 runs under the locality plan made from iteration k's "
                         "coordinates, whose ordering step rides on iteration k's BA (extra workgroups of the first solver launch: "
                         "cuda_ba.forward_delta(..., plan_next=...)); a plan only decides which edges run together.  0: every step orders its own plan "
                         "before its lookup (rounds 1-5; reported as field plan_in_line either way)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-allcores", action="store_true", default=True,
                    help="(default since round 6) cpu_baseline also times the BA with one torch thread per host core (field ba_ms_allcores: ~25 s of "
                         "oversubscribed CPU on a 256-core host, 1000x slower than 16 threads)")
    ap.add_argument("--no-cpu-allcores", dest="cpu_allcores", action="store_false", help="skip the all-cores BA figure of cpu_baseline")
    ap.add_argument("--with-stress", action="store_true",
                    help="also run BASELINE configuration 5 (M=256, n=32, r=5, 1280x720) and add its step rate and lookup roofline as field \"stress\"")
    ap.add_argument("--kernel-reps", type=int, default=50, help="launch pairs timed for the roofline figure")
    ap.add_argument("--mode", default="update-op", choices=["update-op", "train"],
                    help="update-op: the headline metric (BASELINE configuration 2, replicas when --gpus > 1); train: BASELINE "
                         "configurations 3 / 4 — one training step per sequence under DistributedDataParallel (devo_amd/training.py)")
    ap.add_argument("--train-iters", type=int, default=18, help="update iterations per training step (DEVO_base.conf: 18)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="debug (1-GPU boxes): all ranks of a --gpus N run use GPU 0 and gloo carries the collectives — exercises the N > 1 code "
                         "path; the numbers are not a scaling measurement")
    ap.add_argument("--ddp-single", action="store_true",
                    help="debug (1-GPU boxes): a --gpus 1 run creates the RCCL process group and wraps the net in DistributedDataParallel anyway — the "
                         "N > 1 code path of --mode train (communicator, bucket hooks, all-reduce over one rank) on the real backend")
    ap.add_argument("--no-f16", action="store_true", help="skip the secondary fp16-storage measurement (field \"f16\")")
    ap.add_argument("--no-train-probe", action="store_true",
                    help="multi-GPU update-op runs also time a few data-parallel training steps (field \"train_dp\": the RCCL gradient "
                         "all-reduce of BASELINE configuration 4); this skips them")
    ap.add_argument("--api", default="fused", choices=["fused", "reference"],
                    help="reference: ONLY the probe of field \"reference_api\" — the reference's own call sequence (devo/devo.py:210-223,308-344: "
                         "pops.transform, two altcorr.corr calls on the NCHW ring + torch.stack, fastba.BA), launched eagerly, no graph, no plan / "
                         "workspace hand-over — printed as the JSON line")
    ap.add_argument("--no-reference-api", action="store_true", help="skip the reference_api field of the default line")
    ap.add_argument("--no-full-iteration", action="store_true",
                    help="skip the extra field full_update_iteration: a FULL DEVO update iteration — the step with the Update operator "
                         "(devo_amd.update, random weights, fp32 and fp16) between lookup and BA, feeding delta / weight to the BA (the "
                         "headline metric excludes the Update MLP, SURVEY 8d)")
    args = ap.parse_args()
    args.fuse_levels = not args.per_level_launches
    return args


def build_inputs(cfg, seed, device, dtype, layout):
    from devo_amd import synth, altcorr
    n, M, H, W, C = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["C"]
    poses = synth.make_poses(n, seed)
    patches, centres = synth.make_patches(n, M, H, W, seed=seed)
    intr = synth.make_intrinsics(n, H, W)
    ii, jj, kk = synth.full_graph(n, M)
    fmap, gmap = synth.make_features(n, M, C, H, W, centres, seed=seed)
    delta, weight = synth.make_update_outputs(len(ii), seed)
    d = dict(poses0=poses.to(device), patches0=patches.to(device), intr=intr.to(device),
             ii=ii.to(device), jj=jj.to(device), kk=kk.to(device), delta=delta.to(device), weight=weight.to(device),
             lmbda=torch.as_tensor([1e-4], device=device))
    f0 = fmap.to(device)
    f1 = synth.pyramid_l1(f0)
    f0, f1, g = f0.to(dtype), f1.to(dtype), gmap.to(device).to(dtype)
    if layout == "cl":
        f0, f1 = altcorr.channels_last(f0), altcorr.channels_last(f1)
    elif layout == "blk8":
        cb = int(os.environ.get("DEVO_BENCH_CB", "8"))                     # experiment switch: channels per block of the "blk8" layout
        f0, f1 = altcorr.channel_blocked(f0, cb), altcorr.channel_blocked(f1, cb)
    d.update(pyramid=[f0, f1], gmap=g.contiguous())
    # the optimised state (poses + patches) lives in ONE buffer so that a step restores it with a single copy
    npose = d["poses0"].numel()
    off = (npose + 63) // 64 * 64
    flat0 = torch.zeros(off + d["patches0"].numel(), device=device)
    flat0[:npose] = d["poses0"].reshape(-1)
    flat0[off:] = d["patches0"].reshape(-1)
    flat = flat0.clone()
    d["state0"], d["state"] = flat0, flat
    d["poses"] = flat[:npose].view_as(d["poses0"])
    d["patches"] = flat[off:].view_as(d["patches0"])
    cpu = dict(poses=poses, patches=patches, intr=intr, ii=ii, jj=jj, kk=kk, delta=delta, weight=weight,
               fmap=fmap, gmap=gmap)
    return d, cpu


PROBE_TIMEOUT_S = float(os.environ.get("DEVO_BENCH_PROBE_TIMEOUT", "90"))   # multi-GPU default runs: the data-parallel training probe may take this long at most


def alg_bytes(cfg, E, esize):
    """Touch-once traffic of the two-level lookup: fmap2 + gmap + coords + ii,jj + output."""
    n, M, H, W, C, R = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["C"], cfg["R"]
    Dm = 2 * R + 1
    tot = 0
    for (h, w) in ((H, W), (H // 4, W // 4)):
        tot += esize * n * C * h * w + esize * n * M * C * 9 + 4 * E * 2 * 9 + 16 * E + esize * E * Dm * Dm * 9
    return tot


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: spawn the N ranks here (the driver's torch.distributed.run launch sets WORLD_SIZE itself)
        if not torch.cuda.is_available() or (torch.cuda.device_count() < args.gpus and not args.share_gpu):
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count() if torch.cuda.is_available() else 0} GPU(s) visible")
        from devo_amd import distributed as D
        D.launch(rank_main, args.gpus, (sys.argv[1:],))
        return
    rank_main(sys.argv[1:])


def rank_main(argv):
    sys.argv = [sys.argv[0]] + list(argv)
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world} (launch with --nproc-per-node {args.gpus})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from devo_amd import distributed as D
    D.init_from_env("gloo" if args.share_gpu else "nccl", device, force=args.ddp_single)
    assert D.world() == world
    if args.mode == "train":
        out = train_mode(args, device, rank, world)
    elif args.api == "reference":
        out = {"metric": "update-op iterations/sec through the reference's own call sequence (eager)", "unit": "it/s", "n_gpus": 1,
               "reference_api": reference_api_probe(args, device, 1234 + rank)}
        out["value"] = out["reference_api"]["f32_package"]["it_per_s"]
    else:
        out = update_op_mode(args, device, rank, world)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def train_mode(args, device, rank, world, steps=None, warmup=None, iters=None, probe=False):
    """BASELINE configurations 3 (N = 1) and 4 (N > 1): one training step = one sequence per rank (batch = N sequences) through
    `iters` update iterations, loss, backward (DDP all-reduces the 13.59 MB gradient bucket over RCCL), clip, AdamW step."""
    from devo_amd import distributed as D, training as T
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    iters = args.train_iters if iters is None else iters
    ddp = world > 1 or getattr(args, "ddp_single", False)
    net, model, opt = T.build_trainer(device, world, ddp=ddp)
    batch = T.make_batch("cfg2_m80", 1234 + rank, device)
    for _ in range(max(warmup, 1)):
        T.train_step(model, opt, batch, iters=iters)
    if probe:
        # the probe follows the update-op phase in the same process: the first steps still pay MIOpen's solver look-ups, allocator growth
        # and DDP's bucket rebuild (round 3 reported 1578 ms for a 105 ms step after ONE warm-up step).  Warm up until two consecutive
        # steps agree within 10 % (at most 8 more), then time `steps` steps one by one and report the MEDIAN.
        def one():
            D.barrier_sync(device)
            t = time.perf_counter()
            l_ = T.train_step(model, opt, batch, iters=iters)
            D.barrier_sync(device)
            return D.max_over_ranks(time.perf_counter() - t, device), l_
        prev, _ = one()
        for _ in range(8):
            cur, _ = one()
            settled = abs(cur - prev) <= 0.1 * max(cur, prev)
            settled = D.max_over_ranks(0.0 if settled else 1.0, device) == 0.0          # the same decision on every rank (collectives inside)
            prev = cur
            if settled:
                break
        samples = []
        for _ in range(max(steps, 3)):
            t, loss = one()
            samples.append(t)
        samples.sort()
        elapsed, steps = samples[len(samples) // 2] * len(samples), len(samples)
    else:
        D.barrier_sync(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = T.train_step(model, opt, batch, iters=iters)
        D.barrier_sync(device)
        elapsed = D.max_over_ranks(time.perf_counter() - t0, device)
    nparam = net.num_parameters()
    res = {"ms_per_step": round(1e3 * elapsed / steps, 3), "sequences_per_s": round(world * steps / elapsed, 4),
           "update_iterations_per_step": iters, "steps": steps, "loss": float(loss),
           "grad_bucket_bytes": 4 * nparam, "parameters": nparam, "collective": (("DDP all-reduce (gloo: --share-gpu debug run)" if getattr(args, "share_gpu", False) else "DDP all-reduce (RCCL)") if ddp else None)}
    if probe:
        return res
    return {"metric": "training sequences/sec (update + BA path, batch = 1 sequence per GPU)", "value": res["sequences_per_s"], "unit": "seq/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg3/4: n=15 frames (5-bin voxel grids 480x640 through both encoders + scorer), M=80 patches/frame, E={batch['E']} edges, {iters} update iterations per step, "
                                   f"corr backward on 20 % of the edges, 2 differentiable GN steps per iteration, AdamW",
                       "parallelism": f"dp{world}"},
            "train": res}


def reference_api_probe(args, device, seed=1234, iters=60, warm=40):
    """What a caller pays who keeps the reference's OWN call sequence (devo/devo.py:210-223 DEVO.corr / DEVO.reproject, :308-344
    DEVO.update without the network): per update iteration
        coords = pops.transform(SE3(poses), patches, intrinsics, ii, jj, kk).permute(0, 1, 4, 2, 3).contiguous()
        corr   = torch.stack([altcorr.corr(gmap, pyramid[0], coords / 1, kk % (M mem), jj % mem, 3),
                              altcorr.corr(gmap, pyramid[1], coords / 4, kk % (M mem), jj % mem, 3)], -1).view(1, E, -1)
        target = coords[..., 1, 1] + delta;   fastba.BA(poses, patches, intrinsics, target, weight, lmbda, ii, jj, kk, 1, n, 2)
    on the reference's storage — NCHW ring buffers of mem = 32 frames, fp16 (MIXED_PRECISION, devo.py:71-83) or fp32 — launched eagerly:
    no HIP graph, no plan or workspace handed from call to call (each altcorr.corr builds its own locality plan; the converted pyramid
    and the transposed patches are cached per tensor version inside cuda_corr, as they would be in DEVO: they change once per frame).
    Two forms of the reprojection: `package` = devo_amd.projective_ops (one fused kernel without autograd), `modules` = the reference's
    composition over the SE3 group ops (projective_ops.py:53-105 on top of the installed lietorch_backends: what an UNMODIFIED checkout
    gets after devo_amd.backends.install()).  Reported: iterations/s with the GPU as the clock, and the host time to enqueue one
    iteration (no synchronisation inside).  40 untimed iterations first: the host runs ahead of the GPU and the caching allocator keeps growing
    its pool (synchronous hipMallocs) until as many iterations' temporaries are in flight as the run-ahead allows."""
    from devo_amd import synth, altcorr, fastba, projective_ops as pops
    from devo_amd.lietorch import SE3
    cfg = synth.workload(args.workload)
    n, M, H, W, C = cfg["n"], cfg["M"], cfg["H"], cfg["W"], cfg["C"]
    mem = 32
    poses = synth.make_poses(n, seed)
    patches, centres = synth.make_patches(n, M, H, W, seed=seed)
    intr = synth.make_intrinsics(n, H, W).to(device)
    ii, jj, kk = [t.to(device) for t in synth.full_graph(n, M)]
    fmap, gmap = synth.make_features(n, M, C, H, W, centres, seed=seed)
    delta, weight = [t.to(device) for t in synth.make_update_outputs(len(ii), seed)]
    lmbda = torch.as_tensor([1e-4], device=device)
    E = ii.numel()
    out = {}
    for dtn, dt in (("f16", torch.float16), ("f32", torch.float32)):
        fmap1_ = torch.zeros(1, mem, C, H, W, dtype=dt, device=device)             # devo.py:80-81
        fmap2_ = torch.zeros(1, mem, C, H // 4, W // 4, dtype=dt, device=device)
        gmap_ = torch.zeros(mem, M, C, 3, 3, dtype=dt, device=device)               # devo.py:77
        f0 = fmap.to(device)
        fmap1_[:, :n] = f0.to(dt)
        fmap2_[:, :n] = synth.pyramid_l1(f0).to(dt)
        gmap_.view(1, mem * M, C, 3, 3)[:, :n * M] = gmap.to(device).to(dt)
        pyramid, gm = (fmap1_, fmap2_), gmap_.view(1, mem * M, C, 3, 3)
        P0, Q0 = poses.to(device), patches.to(device)
        P, Q = P0.clone(), Q0.clone()
        for form in ("package", "modules"):
            def update():
                P.copy_(P0)
                Q.copy_(Q0)                                                         # (bench harness: fresh state, as in the fused step)
                coords = pops.transform(SE3(P), Q, intr, ii, jj, kk, fused=(form == "package"))
                coords = coords.permute(0, 1, 4, 2, 3).contiguous()
                ii1 = kk % (M * mem)
                jj1 = jj % mem
                corr1 = altcorr.corr(gm, pyramid[0], coords / 1, ii1, jj1, 3)
                corr2 = altcorr.corr(gm, pyramid[1], coords / 4, ii1, jj1, 3)
                corr = torch.stack([corr1, corr2], -1).view(1, E, -1)
                target = coords[..., 1, 1] + delta.float()
                fastba.BA(P, Q, intr, target, weight, lmbda, ii, jj, kk, 1, n, 2)
                return corr
            with torch.no_grad():
                for _ in range(warm):
                    update()
                torch.cuda.synchronize(device)
                t_all, t_host = float("inf"), 0.0
                for _ in range(3):                                                   # the best of three blocks: allocator growth (synchronous
                    t0 = time.perf_counter()                                         # hipMallocs) can still fall into one
                    for _ in range(iters):
                        update()
                    th = time.perf_counter() - t0
                    torch.cuda.synchronize(device)
                    ta = time.perf_counter() - t0
                    if ta < t_all:
                        t_all, t_host = ta, th
            out[f"{dtn}_{form}"] = {"it_per_s": round(iters / t_all, 1), "ms_per_iter": round(1e3 * t_all / iters, 4),
                                    "host_us_per_iter": round(1e6 * t_host / iters, 1)}
        # ---- steady state of the unchanged devo.py (verdict r05 item 2c): every iteration first writes ONE slot of the three rings, the way
        # devo.py:523-527 does for a new frame (`gmap_[k] = ...`, `fmap1_[:, k] = ...`, `fmap2_[:, k] = ...`), then runs the update of the
        # config/default.yaml graph — 45 312 edges of the sliding window (PATCH_LIFETIME 13, REMOVAL_WINDOW 22) in DEVO's order, ring indices
        # modulo mem, t0 = n - OPTIMIZATION_WINDOW.  With the __setitem__ wrapper of devo_amd.backends.ring the lookup converts the written
        # slot only; `whole_ring` = the wrapper switched off: the three tensors are converted again for every frame (rounds 1-5).
        try:
            from devo_amd.backends import ring
            import devo_amd.backends as B_
            nk = 40
            sii, sjj, skk = [t.to(device) for t in synth.sliding_window_graph(nk, M)]
            sE = sii.numel()
            sposes = synth.make_poses(48, seed, trans_step=0.01, rot_step=0.002).to(device)      # overlapping frames (tests/test_gpu_steady_state.py)
            spatches = synth.make_patches(48, M, H, W, seed=seed)[0].to(device)
            sintr = synth.make_intrinsics(48, H, W).to(device)
            sdelta, sweight = [t.to(device) for t in synth.make_update_outputs(sE, seed, sigma=0.5)]
            fm_new = [f0[:, k % n].to(dt).clone() for k in range(4)]          # "new frames" to write (contents do not matter to the clock)
            f1_new = [synth.pyramid_l1(f0[:, k % n][:, None])[:, 0].to(dt).clone() for k in range(4)]
            gm_new = [gmap.to(device).view(n, M, C, 3, 3)[k % n].to(dt).clone() for k in range(4)]
            P1, Q1 = sposes.clone(), spatches.clone()
            state = {"f": nk}

            g_ii, g_jj, g_kk = sii, sjj, skk                                   # the graph's index tensors; every frame gets NEW ones (below)

            def frame_and_update():
                # devo.py:228-231,304-306 rebuild ii / jj / kk with torch.cat / boolean masks for every frame: fresh tensors, so that nothing
                # keyed on the graph (the BA's remembered index tables, the Update operator's neighbour tables) survives from frame to frame
                sii, sjj, skk = g_ii.clone(), g_jj.clone(), g_kk.clone()
                k = state["f"] % mem
                state["f"] += 1
                gmap_[k] = gm_new[k % 4]                                      # devo.py:524
                fmap1_[:, k] = fm_new[k % 4]                                  # devo.py:526
                fmap2_[:, k] = f1_new[k % 4]                                  # devo.py:527
                P1.copy_(sposes); Q1.copy_(spatches)
                coords = pops.transform(SE3(P1), Q1, sintr, sii, sjj, skk, fused=True).permute(0, 1, 4, 2, 3).contiguous()
                ii1, jj1 = skk % (M * mem), sjj % mem
                corr = torch.stack([altcorr.corr(gm, pyramid[0], coords / 1, ii1, jj1, 3), altcorr.corr(gm, pyramid[1], coords / 4, ii1, jj1, 3)], -1).view(1, sE, -1)
                target = coords[..., 1, 1] + sdelta
                fastba.BA(P1, Q1, sintr, target, sweight, lmbda, sii, sjj, skk, nk - 10, nk, 2)
                return corr
            for label, track in (("per_slot", True), ("whole_ring", False)):
                on = ring.track_ring_writes(track)                             # (both bindings keep write records since round 6)
                with torch.no_grad():
                    for _ in range(10):
                        frame_and_update()
                    torch.cuda.synchronize(device)
                    best = float("inf")
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(30):
                            frame_and_update()
                        torch.cuda.synchronize(device)
                        best = min(best, time.perf_counter() - t0)
                out[f"{dtn}_steady_state_{label}"] = {"it_per_s": round(30 / best, 1), "ms_per_iter": round(1e3 * best / 30, 4), "edges": sE, "write_tracking": bool(on)}
            if dtn == "f16":
                # ... and the same frame with the Update operator in its place (devo.py:308-317: fp32 parameters under autocast, fp16 rings,
                # `ctx = imap[:, kk % (M mem)]`, the recurrent state fed back): what ONE frame of the unchanged devo.py costs on this path
                # behind the patchifier (whose encoders are MIOpen's)
                from devo_amd.update import Update
                ring.track_ring_writes(True)
                torch.manual_seed(seed)
                upd = Update(3).to(device).eval()
                with torch.no_grad():
                    for p_ in upd.parameters():
                        if p_.dim() == 2 and p_.shape[0] == 2:
                            p_.mul_(0.05)                                     # small flow updates: the adjustment stays in its basin
                imap_ = (torch.randn(mem, M, 384, device=device) * 0.5).to(dt)
                st2 = {"f": nk, "net": torch.zeros(1, sE, 384, device=device, dtype=dt)}

                def frame_full(fused_lookup=False):
                    sii, sjj, skk = g_ii.clone(), g_jj.clone(), g_kk.clone()   # (a new graph per frame, as above)
                    ring_idx = skk % (M * mem)
                    k = st2["f"] % mem
                    st2["f"] += 1
                    gmap_[k] = gm_new[k % 4]; fmap1_[:, k] = fm_new[k % 4]; fmap2_[:, k] = f1_new[k % 4]
                    P1.copy_(sposes); Q1.copy_(spatches)
                    coords = pops.transform(SE3(P1), Q1, sintr, sii, sjj, skk, fused=True).permute(0, 1, 4, 2, 3).contiguous()
                    with torch.autocast("cuda", enabled=True, dtype=torch.float16):
                        ii1, jj1 = ring_idx, sjj % mem
                        if fused_lookup:                                      # DEVO.corr's three lines (devo.py:215-217) as one call
                            corr = altcorr.corr_pyramid(gm, pyramid, coords, ii1, jj1, 3)
                        else:
                            corr = torch.stack([altcorr.corr(gm, pyramid[0], coords / 1, ii1, jj1, 3), altcorr.corr(gm, pyramid[1], coords / 4, ii1, jj1, 3)], -1).view(1, sE, -1)
                        ctx = imap_.view(1, mem * M, 384)[:, ring_idx]
                        st2["net"], (delta_, weight_, _) = upd(st2["net"], ctx, corr, None, sii, sjj, skk)
                    target = coords[..., 1, 1] + delta_.float()
                    fastba.BA(P1, Q1, sintr, target, weight_.float(), lmbda, sii, sjj, skk, nk - 10, nk, 2)
                with torch.no_grad():
                    for _ in range(10):
                        frame_full()
                    torch.cuda.synchronize(device)
                    best, best_host = float("inf"), 0.0
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(30):
                            frame_full()
                        th = time.perf_counter() - t0
                        torch.cuda.synchronize(device)
                        if time.perf_counter() - t0 < best:
                            best, best_host = time.perf_counter() - t0, th
                out["f16_steady_state_frame_with_update_operator"] = {"frames_per_s": round(30 / best, 1), "ms_per_frame": round(1e3 * best / 30, 4), "edges": sE,
                                                                      "host_ms_per_frame": round(1e3 * best_host / 30, 4),
                                                                      "note": "one ring slot written + reproject + two-level lookup + Update operator (fp32 parameters under autocast) + 2 GN "
                                                                              "iterations on its outputs, eager, the reference's call sequence"}
                with torch.no_grad():                                        # ... and with DEVO.corr's two calls + torch.stack as ONE altcorr.corr_pyramid call
                    st2["net"] = torch.zeros(1, sE, 384, device=device, dtype=dt)
                    for _ in range(10):
                        frame_full(True)
                    torch.cuda.synchronize(device)
                    best = float("inf")
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(30):
                            frame_full(True)
                        torch.cuda.synchronize(device)
                        best = min(best, time.perf_counter() - t0)
                # ... and the WHOLE per-frame work of devo.py's __call__ on this path: the Patchifier on one 480 x 640 event frame (devo.py:250, under autocast),
                # its outputs written into the rings as devo.py:520-527 does (two avg_pool2d calls among them), then the frame above (unchanged call sequence)
                try:
                    from devo_amd.patchifier import Patchifier
                    torch.manual_seed(seed)
                    pfm = Patchifier().to(device).eval()
                    ev = torch.randn(1, 1, 5, 4 * H, 4 * W, device=device)

                    def whole_frame():
                        with torch.autocast("cuda", enabled=True, dtype=torch.float16):
                            fm, gmp, imp, pts, _ = pfm(ev, M)
                        k = st2["f"] % mem
                        imap_[k] = imp.squeeze()
                        gmap_[k] = gmp.squeeze()
                        fmap1_[:, k] = torch.nn.functional.avg_pool2d(fm[0], 1, 1)
                        fmap2_[:, k] = torch.nn.functional.avg_pool2d(fm[0], 4, 4)
                        frame_full(False)
                    with torch.no_grad():
                        st2["net"] = torch.zeros(1, sE, 384, device=device, dtype=dt)
                        for _ in range(10):
                            whole_frame()
                        torch.cuda.synchronize(device)
                        best_w = float("inf")
                        for _ in range(3):
                            t0 = time.perf_counter()
                            for _ in range(30):
                                whole_frame()
                            torch.cuda.synchronize(device)
                            best_w = min(best_w, time.perf_counter() - t0)
                    out["f16_steady_state_whole_frame_with_patchifier"] = {
                        "frames_per_s": round(30 / best_w, 1), "ms_per_frame": round(1e3 * best_w / 30, 4), "edges": sE,
                        "note": "the Patchifier on one 480 x 640 event frame under autocast (devo.py:250) + its outputs into the rings (devo.py:520-527) + the frame "
                                "with the Update operator above: what devo.py's __call__ enqueues per frame on this path, keyframe bookkeeping aside"}
                    del pfm
                except Exception as ex:                              # noqa: BLE001 — an extra field
                    out["f16_steady_state_whole_frame_error"] = f"{type(ex).__name__}: {ex}"[:300]
                out["f16_steady_state_frame_fused_lookup"] = {"frames_per_s": round(30 / best, 1), "ms_per_frame": round(1e3 * best / 30, 4), "edges": sE,
                                                              "note": "the same frame with devo.py:215-217 (two altcorr.corr calls + torch.stack) replaced by altcorr.corr_pyramid "
                                                                      "(INTEGRATION.md section 2): both levels in one launch, written straight into the stacked layout"}
                del upd, imap_
            ring.track_ring_writes(False)
            del P1, Q1, sposes, spatches
        except Exception as ex:                                      # noqa: BLE001 — an extra field must not cost the probe
            out[f"{dtn}_steady_state_error"] = f"{type(ex).__name__}: {ex}"[:300]
        del fmap1_, fmap2_, gmap_
    out["steady_state"] = ("one new frame per iteration written into ONE slot of the fp16 / fp32 rings as devo.py:523-527 does, then the reference's call sequence on the "
                           "config/default.yaml graph (45 312 edges of the sliding window after 40 keyframes, indices modulo the ring, 10 optimised poses): "
                           "per_slot = devo_amd.backends.ring records the writes and the lookup converts the written slot; whole_ring = every frame converts the rings again")
    out["what"] = ("devo.py:210-223,308-344 as written (transform, permute, two altcorr.corr on the NCHW ring of 32 frames + torch.stack, "
                   "target, fastba.BA with 2 GN iterations), eager launches, caches warm; package = devo_amd.projective_ops (fused "
                   "reprojection), modules = the reference's SE3 group-op composition over lietorch_backends")
    return out


def update_op_mode(args, device, rank, world, dtype_name=None, secondary=False):
    dtn = dtype_name or args.dtype
    from devo_amd import synth, altcorr, distributed as D
    from devo_amd.backends import cuda_ba, cuda_corr

    cfg = synth.workload(args.workload)
    dtype = torch.float32 if dtn == "f32" else torch.float16
    d, cpu = build_inputs(cfg, 1234 + rank, device, dtype, args.layout)
    n, M, R = cfg["n"], cfg["M"], cfg["R"]
    E = d["ii"].numel()
    Np = d["patches"].shape[1]
    ws = cuda_ba.workspace(E, Np, n - 1, device)
    Dm = 2 * R + 1
    corr_out = torch.empty(1, E, Dm * Dm * 9 * 2, dtype=dtype, device=device)

    def lookup(coords, order=None):
        if not args.fuse_levels:
            if order is None:
                order = cuda_corr.plan(coords, d["jj"], n, cfg["H"], radius=R, width=cfg["W"], l1=PL1)   # locality plan, shared by both levels
            for lvl, (fm, s) in enumerate(zip(d["pyramid"], (1, 4))):
                cuda_corr.forward_into(corr_out, d["gmap"], fm, coords, d["kk"], d["jj"], R, Dm * Dm * 18, 2, lvl, order=order,
                                       coord_div=float(s))                   # the kernel looks up at coords / s
        else:
            # plan + ONE launch for both levels (workgroups of the two levels alternate on every CU)
            cuda_corr.forward_pyramid(d["gmap"], d["pyramid"], coords, d["kk"], d["jj"], R, (1, 4), out=corr_out, order=order)

    PL1 = 4 if (args.plan == "groups" and args.fuse_levels) else 0
    prep_stream = torch.cuda.Stream() if args.overlap_prepare else None
    probe = {"on": False, "ev": []}          # eager steps after the timed region record HIP events around the lookup launch(es)

    def lookup_probed(coords, order=None):
        if not probe["on"]:
            return lookup(coords, order=order)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lookup(coords, order=order); b.record()
        probe["ev"].append((a, b))

    plan_prev = [None]                                                   # the finished plan of the previous step (--plan-lag 1)

    def step(k=0, every=None, lag=None):
        cur = torch.cuda.current_stream()
        fresh_graph = (k % max(1, every or args.prepare_every)) == 0      # this step sees a "new" patch graph: the index tables are rebuilt
        lag_on = bool(args.plan_lag if lag is None else lag) and not (args.separate_target or args.separate_index_kernels) and prep_stream is None
        lagged = lag_on and not fresh_graph and plan_prev[0] is not None   # this step's lookup runs under the previous step's plan
        if prep_stream is not None and fresh_graph:
            # the index half of the BA (unique patches, edges grouped by patch) depends on kk only: it runs on a second
            # stream under the reprojection + lookup (a fork/join inside the captured graph)
            prep_stream.wait_stream(cur)
            with torch.cuda.stream(prep_stream):
                cuda_ba.prepare(d["kk"], Np, n - 1, ws)
        torch.mul(d["state0"], 1.0, out=d["state"])                        # fresh poses + patches (bench harness; an elementwise kernel: rocclr's copyBuffer takes 5 us)
        # reprojection; the kernel also emits the lookup's plan bins while it holds the coordinates
        coords, order = cuda_ba.transform(d["poses"], d["patches"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp",
                                          plan_for=(n, cfg["H"], R, cfg["W"], PL1))
        own_plan = order                                                   # this step's plan buffer (bins written, not yet ordered)
        if lagged:
            order = plan_prev[0]                                           # (own_plan is ordered by this step's BA, for the next step)
        elif args.separate_index_kernels or prep_stream is not None or not fresh_graph:
            order = cuda_corr.plan_finish(order, d["jj"], n, cfg["H"], R, width=cfg["W"], l1=PL1)
            if not fresh_graph:
                pass                                                       # the workspace holds this kk's tables already (cuda_ba.forward's cache, here explicit)
            elif prep_stream is None:
                cuda_ba.prepare(d["kk"], Np, n - 1, ws)
        else:
            # the plan's ordering step and the BA's index preparation (independent of each other) in ONE launch
            cuda_ba.prepare(d["kk"], Np, n - 1, ws, plan=(order, n, cfg["H"], cfg["W"], PL1))
        lookup_probed(coords, order=order)
        if prep_stream is not None and fresh_graph:
            cur.wait_stream(prep_stream)
        if not (args.separate_target or args.separate_index_kernels):
            # devo.py:330 (target = centre of the reprojected patch + delta) formed inside the BA: same fp32 addition
            cuda_ba.forward_delta(d["poses"], d["patches"], d["intr"], coords, d["delta"], d["weight"], d["lmbda"],
                                  d["ii"], d["jj"], d["kk"], 1, n, 2, ws, plan_next=(own_plan, n, cfg["H"], cfg["W"], PL1) if lagged else None)
            plan_prev[0] = own_plan if lag_on else None
            return
        target = coords[:, :, :, 1, 1] + d["delta"]                        # devo.py:330
        cuda_ba.forward(d["poses"], d["patches"], d["intr"], target, d["weight"], d["lmbda"],
                        d["ii"], d["jj"], d["kk"], 1, n, 2, ws=ws, prepared=True)

    # ---- warm up eagerly once (library load, kernel code upload), then capture
    step()
    torch.cuda.synchronize()
    gevery = ginline = None
    if args.no_graph:
        count = [0]

        def run():
            step(count[0])
            count[0] += 1
    else:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side):
                step()
            gmany = None
            if args.steps_per_graph > 1:                                   # several consecutive steps in one graph launch
                gmany = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gmany, stream=side):
                    for k in range(args.steps_per_graph):
                        step(k)
                if args.prepare_every > 1 and not secondary:               # the same steps with a NEW patch graph in every one of them
                    gevery = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gevery, stream=side):
                        for k in range(args.steps_per_graph):
                            step(k, every=1)
                    if args.plan_lag:                                      # ... and with every step ordering its own plan in front of its lookup
                        ginline = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(ginline, stream=side):
                            for k in range(args.steps_per_graph):
                                step(k, lag=0)
        torch.cuda.current_stream().wait_stream(side)
        run = graph.replay
    for _ in range(args.warmup):
        run()
    n_many = 0
    if not args.no_graph and gmany is not None:
        n_many = args.steps // args.steps_per_graph
        gmany.replay()

    def timed_region():
        D.barrier_sync(device)
        t0 = time.perf_counter()
        for _ in range(n_many):
            gmany.replay()
        for _ in range(args.steps - n_many * args.steps_per_graph):       # EXACTLY --steps steps
            run()
        D.barrier_sync(device)
        return D.max_over_ranks(time.perf_counter() - t0, device)

    # The contract's timed region: exactly --steps steps between barrier + synchronize pairs, MAX over ranks.  A short region
    # (20 steps are 6 ms: two graph launches; the default 200 steps 45 ms) is dominated by launch / synchronisation jitter and, as the
    # first thing a fresh box runs, by the clocks still ramping up (54.7 ms seen where every later region takes 45.2), so regions below
    # 0.25 s are measured several times and the MEDIAN region is reported ("timed_regions" in the JSON line says how many).
    elapsed = timed_region()
    regions = 1
    if elapsed < 0.25:
        regions = int(min(25, max(5, 0.25 / max(elapsed, 1e-4)))) | 1
        if world > 1:
            regions = int(D.max_over_ranks(float(regions), device))          # the same count on every rank (collectives inside)
        samples = sorted([elapsed] + [timed_region() for _ in range(regions - 1)])
        elapsed = samples[len(samples) // 2]
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed
    ms_every = None
    if gevery is not None:                                                 # every step rebuilds the BA's index tables (rounds 1-5's step)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gevery.replay()
        ts = []
        for _ in range(5):
            e0.record(); gevery.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.steps_per_graph)
        ms_every = sorted(ts)[2]
    ms_lag_ev = None
    if ginline is not None and gmany is not None:                          # the timed steps by the same method as the two variants below (events around graph replays)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gmany.replay()
        ts = []
        for _ in range(5):
            e0.record(); gmany.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.steps_per_graph)
        ms_lag_ev = sorted(ts)[2]
    ms_inline = None
    if ginline is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ginline.replay()
        ts = []
        for _ in range(5):
            e0.record(); ginline.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.steps_per_graph)
        ms_inline = sorted(ts)[2]

    # ---- roofline figure for the dominant kernel (altcorr lookup), HIP events on the launch stream
    coords = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
    for _ in range(5):
        lookup(coords)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # only the lookup kernels sit between the events
    order = cuda_corr.plan(coords, d["jj"], n, cfg["H"], radius=R, width=cfg["W"], l1=PL1)
    torch.cuda.synchronize()

    def lookups():
        for _ in range(args.kernel_reps):
            lookup(coords, order=order)
    if args.no_graph:
        ev0.record(); lookups(); ev1.record()
    else:
        # replayed from a HIP graph like the timed step: eager back-to-back launches add 5-10 us of launch gap per kernel,
        # which rocprofv3's kernel durations (profiles/) do not contain
        kgraph = torch.cuda.CUDAGraph()
        kside = torch.cuda.Stream()
        kside.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(kside):
            with torch.cuda.graph(kgraph, stream=kside):
                lookups()
        torch.cuda.current_stream().wait_stream(kside)
        kgraph.replay()
        torch.cuda.synchronize()
        ev0.record(); kgraph.replay(); ev1.record()
    torch.cuda.synchronize()
    launches = (1 if args.fuse_levels else 2) * args.kernel_reps
    t_back_to_back = ev0.elapsed_time(ev1) * 1e-3 / launches                # s per launch, lookups only, back to back
    # ... and the same launch where the timed region runs it: inside the step, between the reprojection / index kernels and the BA
    # (HIP events around the lookup of eager steps).  This is the figure rocprofv3's per-kernel average of this command is made of
    # (the step's launches outnumber the ones above), and it is the lower one: a lookup that follows another lookup starts while
    # the 96 MB output of its predecessor is still draining.
    probe["on"] = True
    for i in range(args.kernel_reps):
        step(i)                                                            # (the timed region's mix: one new patch graph per prepare_every steps)
    torch.cuda.synchronize()
    probe["on"] = False
    t_in_step = sum(a.elapsed_time(b) for a, b in probe["ev"]) * 1e-3 / max(1, len(probe["ev"])) / (1 if args.fuse_levels else 2)
    # what an event pair costs by itself (nothing between the two records), measured the same way and subtracted
    empty = []
    for _ in range(args.kernel_reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        step_prefix = cuda_ba.transform(d["poses0"], d["patches0"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")   # (a kernel in front, as in the step)
        a.record(); b.record()
        empty.append((a, b))
    torch.cuda.synchronize()
    t_event_pair = sum(a.elapsed_time(b) for a, b in empty) * 1e-3 / len(empty)
    # roofline figure: the back-to-back launches replayed from a graph (no event-pair correction; rocprofv3's per-kernel average of this
    # command, profiles/, lies between this and the raw in-step figure)
    # round 6 (verdict item 4): the line leads with what the STEP pays — the lookup between the reprojection / ordering kernels and the
    # BA, cold L2 behind them (events around the launch in eager steps, nothing subtracted; rocprofv3's per-kernel average of this command
    # is made of these launches) — and carries the warm back-to-back figure as a secondary key
    t_launch = t_in_step
    b_alg = alg_bytes(cfg, E, 4 if dtype == torch.float32 else 2) / (1.0 if args.fuse_levels else 2.0)   # bytes per launch
    achieved = b_alg / t_launch / 1e9
    f_alg = 2.0 * cfg["C"] * E * 9 * (2 * R + 2) ** 2 * (2.0 if args.fuse_levels else 1.0)    # flops per launch

    # BA alone (2 Gauss-Newton iterations), for the ">= 10x the CPU ba.py solve" target
    tgt = coords[:, :, :, 1, 1] + d["delta"]
    ev0.record()
    for _ in range(20):
        d["state"].copy_(d["state0"])
        cuda_ba.forward(d["poses"], d["patches"], d["intr"], tgt, d["weight"], d["lmbda"], d["ii"], d["jj"], d["kk"], 1, n, 2, ws=ws)
    ev1.record()
    torch.cuda.synchronize()
    t_ba_gpu = ev0.elapsed_time(ev1) * 1e-3 / 20                           # unchanged kk: the index tables of the first call serve the others
    ev0.record()
    for _ in range(20):
        d["state"].copy_(d["state0"])
        cuda_ba.prep_invalidate()                                          # a new patch graph every call: the index tables are rebuilt
        cuda_ba.forward(d["poses"], d["patches"], d["intr"], tgt, d["weight"], d["lmbda"], d["ii"], d["jj"], d["kk"], 1, n, 2, ws=ws)
    ev1.record()
    torch.cuda.synchronize()
    t_ba_gpu_new = ev0.elapsed_time(ev1) * 1e-3 / 20

    # which lookup kernel the library picks for this configuration (devo_amd/csrc/corr.hip: launch_staged)
    mfma = cfg["C"] == 128 and os.environ.get("DEVO_CORR_MFMA", "1")[:1] != "0"          # fp32 and fp16 storage
    # (an NCHW pyramid of >= 1024 edges goes through cuda_corr's cached channel-blocked copy, i.e. the same fast kernel)
    nchw_direct = args.layout == "nchw" and (E < 1024 or os.environ.get("DEVO_CORR_NCHW_DIRECT", "0") == "1" or cfg["C"] % 8 != 0)
    lookup_kernel = "corr_fwd_generic_kernel" if nchw_direct else ("corr_fwd_mfma_kernel" if mfma else "corr_fwd_cl_kernel")
    if lookup_kernel == "corr_fwd_mfma_kernel" and cuda_corr.MM_KERNEL:
        lookup_kernel = "corr_fwd_mm_kernel"                        # dense-product kernel (corr_mm.h): fused lookups and per-level launches alike
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            recs = json.load(f)
        key = f"{args.workload}/{dtn}/{'blk8' if args.layout == 'nchw' and not nchw_direct else args.layout}"
        rec = (recs.get(key + "/staged") if lookup_kernel == "corr_fwd_cl_kernel" else None) or recs.get(key)
        per_launch = 2 if args.fuse_levels else 1                  # records are per level launch ...
        if args.fuse_levels and lookup_kernel == "corr_fwd_mfma_kernel" and recs.get(key + "/fused"):
            rec, per_launch = recs[key + "/fused"], 1              # ... except the two-level launch's own record
        if args.fuse_levels and lookup_kernel == "corr_fwd_mm_kernel" and recs.get(key + "/fused/mm"):
            rec, per_launch = recs[key + "/fused/mm"], 1           # the dense-product kernel's two-level launch
        if rec and rec.get("kernel", "corr_fwd_cl_kernel") == lookup_kernel:
            traffic = int((2.0 * rec["FETCH_SIZE_KB"] + rec["WRITE_SIZE_KB"]) * 1024) * per_launch
            traffic_src = f"profiles/pmc_traffic.json ({rec['round']}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 2x read correction)"
    except (OSError, ValueError, KeyError):
        pass

    out = {
        "metric": "update-op iterations/sec (altcorr+fastba) at 96 patches, N=15 keyframes",
        "value": round(value, 2), "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "timed_regions": regions, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtn, "data": "synthetic",
        "config": {"workload": f"{args.workload}: M={M} patches/frame, n={n} keyframes, E={E} edges, r={R}, "
                               f"2 pyramid levels {cfg['H']}x{cfg['W']} + /4, C={cfg['C']}, 2 GN iterations, "
                               f"pyramid layout {args.layout}, {'both levels in one lookup launch' if args.fuse_levels else 'one lookup launch per level'}, "
                               f"{('HIP graph, ' + str(args.steps_per_graph) + ' step(s) per graph launch') if not args.no_graph else 'eager'}"
                               f"{', BA index preparation on a second stream' if args.overlap_prepare else ''}",
                   "parallelism": f"replicas x{world}" + (" (debug: all ranks on ONE GPU, --share-gpu)" if args.share_gpu else "")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": lookup_kernel,
                     # the FMA ceiling beside the HBM figure.  F_alg = 2 * C * E * 9 * D^2 per level; fp32 peak
                     # 157.3 TFLOP/s (vector FMA and fp32 MFMA alike); t_min = max(B_alg / HBM peak, F_alg / fp32 peak)
                     "alg_gflop_per_launch": round(f_alg / 1e9, 3), "achieved_tflops": round(f_alg / t_launch / 1e12, 2),
                     "t_min_us": round(max(b_alg / (HBM_PEAK_GBS * 1e9), f_alg / 157.3e12) * 1e6, 1),
                     "alg_bytes_per_launch": int(b_alg), "us_per_launch": round(t_launch * 1e6, 2),
                     "us_per_launch_back_to_back": round(t_back_to_back * 1e6, 2), "us_per_launch_in_step_raw": round(t_in_step * 1e6, 2),
                     "event_pair_us": round(t_event_pair * 1e6, 2),
                     "frac_back_to_back": round(b_alg / t_back_to_back / 1e9 / HBM_PEAK_GBS, 4),
                     "timing": "achieved / frac / us_per_launch: HIP events around the lookup launch of " + str(args.kernel_reps) + " eager steps (the kernel where the step runs "
                               "it, nothing subtracted: the pair of events itself costs event_pair_us); back_to_back: events around " + str(args.kernel_reps) +
                               " lookup launches replayed back to back from a HIP graph"},
        "ba": dict({"gpu_ms": round(t_ba_gpu * 1e3, 4), "gpu_ms_new_graph": round(t_ba_gpu_new * 1e3, 4),
                    "note": "gpu_ms / launches / kernels: cuda_ba.forward (2 GN iterations) on an unchanged kk — the index tables (unique patches, edges grouped by patch) "
                            "come from forward()'s prepared-table cache; gpu_ms_new_graph: every call rebuilds them (one more launch)"},
                   **ba_kernel_report(run_ba=lambda: cuda_ba.forward(
                       d["poses"], d["patches"], d["intr"], tgt, d["weight"], d["lmbda"], d["ii"], d["jj"], d["kk"], 1, n, 2, ws=ws))),
    }
    out["config"]["ba_index_tables"] = f"rebuilt every {args.prepare_every} steps (one patch graph = {args.prepare_every} update iterations)"
    out["config"]["lookup_plan"] = ("on an unchanged patch graph the lookup runs under the plan made from the previous iteration's coordinates; its ordering "
                                    "step rides on the BA's first solver launch (cuda_ba.forward_delta(..., plan_next=...)); a new patch graph orders its plan in line"
                                    if (args.plan_lag and not (args.separate_target or args.separate_index_kernels or args.overlap_prepare))
                                    else "ordered in line by every step, in front of its lookup")
    if ms_inline is not None:
        out["plan_in_line"] = {"ms_per_step": round(ms_inline, 4), "value": round(world * 1e3 / ms_inline, 2), "unit": "it/s",
                               "same_method_with_the_lagged_plan": {"ms_per_step": round(ms_lag_ev, 4), "value": round(world * 1e3 / ms_lag_ev, 2),
                                                                     "method": "HIP events around one replay of the 18-step graph, median of 5 (the headline `value` is "
                                                                               "wall clock over the whole timed region, graph launches and the closing synchronisation included)"},
                               "note": "the same steps with --plan-lag 0: every step orders its own locality plan (corr_order_kernel, ~9 us) between the "
                                       "reprojection and its lookup (rounds 1-5's step)"}
    if ms_every is not None:
        out["new_graph_every_step"] = {"ms_per_step": round(ms_every, 4), "value": round(world * 1e3 / ms_every, 2), "unit": "it/s",
                                       "note": "the same steps with the BA's index tables rebuilt in EVERY step (rounds 1-5's step; DEVO's steady-state inference "
                                               "appends edges once per frame: one update() per new graph)"}

    if not args.no_full_iteration and args.workload == "cfg2" and world == 1:
        try:                                                         # an extra field must not cost the line
            # A FULL update iteration as devo.py:305-340 runs it: reprojection, lookup, the Update operator (devo_amd.update, random weights)
            # on the lookup's output, target = centre + delta, 2 GN iterations with the predicted weights.  An extra field: the headline
            # metric excludes the Update MLP.  fp32 = every Linear layer on csrc/linear.hip's split-precision GEMM.
            from devo_amd.update import Update
            full = {}
            # (the fp16-storage pass of the bench — DEVO's inference precision: fp16 pyramid, fp16 operator, fp32 BA — times the fp16 operator only)
            for udt, key in (((torch.float16, "f16"),) if secondary else ((torch.float32, "f32"), (torch.float16, "f16"))):
                torch.manual_seed(1234 + rank)
                upd = Update(3).to(device).to(udt).eval()
                net_h = torch.zeros(1, E, 384, device=device, dtype=udt)
                inp_h = torch.randn(1, E, 384, device=device, dtype=udt) * 0.1

                def full_iteration():
                    torch.mul(d["state0"], 1.0, out=d["state"])
                    coords = cuda_ba.transform(d["poses"], d["patches"], d["intr"], d["ii"], d["jj"], d["kk"], layout="2pp")
                    lookup(coords)
                    with torch.no_grad():
                        _, (delta, weight, _) = upd(net_h, inp_h, corr_out.to(udt), None, d["ii"], d["jj"], d["kk"])
                    target = coords[:, :, :, 1, 1] + delta.float()
                    cuda_ba.forward(d["poses"], d["patches"], d["intr"], target, weight.float(), d["lmbda"],
                                    d["ii"], d["jj"], d["kk"], 1, n, 2, ws=ws)
                for _ in range(3):
                    full_iteration()
                torch.cuda.synchronize()
                run_full, how = full_iteration, "eager launches"
                if not args.no_graph:                                      # the Update operator's group tables are cached by now: no host sync left
                    g2 = torch.cuda.CUDAGraph()
                    s2 = torch.cuda.Stream()
                    s2.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s2):
                        with torch.cuda.graph(g2, stream=s2):
                            full_iteration()
                    torch.cuda.current_stream().wait_stream(s2)
                    run_full, how = g2.replay, "HIP graph"
                for _ in range(3):
                    run_full()
                torch.cuda.synchronize()
                ev0.record()
                for _ in range(50):
                    run_full()
                ev1.record()
                torch.cuda.synchronize()
                full[key + "_ms"] = round(ev0.elapsed_time(ev1) / 50, 4)
                if key == "f16":
                    # the Update operator alone (verdict r05 item 4): its GEMMs are the one MFMA-roofline part of the path — 16 products of
                    # E x 384 x 384, the 882-wide first layer of the correlation branch and the two heads
                    c16 = corr_out.to(udt)
                    with torch.no_grad():
                        for _ in range(3):
                            upd(net_h, inp_h, c16, None, d["ii"], d["jj"], d["kk"])
                        ev0.record()
                        for _ in range(20):
                            upd(net_h, inp_h, c16, None, d["ii"], d["jj"], d["kk"])
                        ev1.record()
                    torch.cuda.synchronize()
                    t_op = ev0.elapsed_time(ev1) * 1e-3 / 20
                    fl = 2.0 * E * (16 * 384 * 384 + 882 * 384 + 2 * 2 * 384)
                    out["update_op"] = {"f16_ms": round(t_op * 1e3, 4), "gflop": round(fl / 1e9, 1), "achieved_tflops": round(fl / t_op / 1e12, 1),
                                        "peak_tflops": 2500.0, "frac": round(fl / t_op / 2.5e15, 4), "bound": "mfma",
                                        "note": "devo_amd.update.Update (fp16 weights and state), eager launches from pointers collected once per parameter version; "
                                                "peak = the dense fp16 MFMA figure of MI355X_MICROARCH.md"}
                    del c16
                del upd
            full["note"] = (f"reproject + 2-level lookup ({dtn} pyramid) + Update operator (fp32 / fp16 weights and state, random weights) + "
                            f"2 GN iterations on its outputs, {how}")
            out["full_update_iteration"] = full
        except Exception as ex:                                      # noqa: BLE001
            out["full_update_iteration"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}

    if secondary:
        return out
    if dtn == "f32" and not args.no_f16:
        # the reference's inference precision (devo/devo.py:71-77: fp16 feature maps under autocast) as a second measurement
        # of the same step; fp32 accumulation in the lookup, so every output is at least as accurate as the reference's
        del d, corr_out
        torch.cuda.empty_cache()
        h = update_op_mode(args, device, rank, world, dtype_name="f16", secondary=True)
        out["f16"] = {"value": h["value"], "unit": "it/s", "ms_per_step": h["ms_per_step"],
                      "roofline": {k: h["roofline"].get(k) for k in ("achieved", "frac", "kernel", "alg_bytes_per_launch", "us_per_launch", "us_per_launch_back_to_back", "traffic")},
                      "note": "same step with fp16-storage feature pyramid + patch features (DEVO's inference precision), fp32 accumulation"}
        if "full_update_iteration" in h:                         # the full iteration at DEVO's inference precision: fp16 pyramid AND fp16 operator
            out["f16"]["full_update_iteration_ms"] = h["full_update_iteration"]["f16_ms"]
    if rank == 0 and world == 1 and not args.no_reference_api and args.workload == "cfg2":
        # what the reference's own (unfused, eager) call sequence costs on the same machine: bench.py --api reference alone prints it
        torch.cuda.empty_cache()
        try:
            out["reference_api"] = reference_api_probe(args, device, 1234 + rank)
            out["reference_api"]["vs_fused_api"] = round(out["reference_api"]["f32_package"]["it_per_s"] / out["value"], 3)
        except Exception as ex:                                  # noqa: BLE001 — an extra field must not cost the line
            out["reference_api"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if world > 1 and not args.no_train_probe:
        # BASELINE configuration 4: data-parallel training steps — the one collective of the path (13.59 MB gradient all-reduce).
        # The probe is an extra: whatever happens inside it (an exception on one rank, a collective that never returns) must not cost
        # the line of the metric measured above.  A rank that raises records the error; a watchdog per rank ends a probe that hangs —
        # rank 0 prints the line first.
        import threading

        def give_up():
            if rank == 0:
                out["train_dp"] = {"error": f"probe did not finish within {PROBE_TIMEOUT_S} s"}
                print(json.dumps(out), flush=True)
            os._exit(0)

        timer = threading.Timer(PROBE_TIMEOUT_S, give_up)
        timer.daemon = True
        timer.start()
        try:
            try:
                del d, corr_out                                  # the update-op phase's buffers (gone already when the fp16 pass ran)
            except NameError:
                pass
            torch.cuda.empty_cache()
            out["train_dp"] = train_mode(args, device, rank, world, steps=3, warmup=1, iters=2, probe=True)
            out["train_dp"]["note"] = ("probe: median of 3 steps of 2 update iterations each, after warm-up steps until two consecutive ones agree "
                                       "within 10 % (bench.py --mode train runs the full 18-iteration step)")
        except Exception as ex:                                  # noqa: BLE001 — reported, not fatal
            out["train_dp"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        timer.cancel()
    if rank == 0 and world == 1 and args.with_stress and args.workload != "stress":
        try:
            torch.cuda.empty_cache()
            sargs = argparse.Namespace(**dict(vars(args), workload="stress", steps=max(10, args.steps // 10), warmup=2, steps_per_graph=1, kernel_reps=10))
            st = update_op_mode(sargs, device, rank, world, dtype_name=dtn, secondary=True)
            out["stress"] = {"value": st["value"], "unit": "it/s", "ms_per_step": st["ms_per_step"], "workload": st["config"]["workload"],
                             "roofline": {k: st["roofline"].get(k) for k in ("achieved", "frac", "kernel", "alg_bytes_per_launch", "us_per_launch", "traffic", "traffic_source")},
                             "ba": st["ba"]}
        except Exception as ex:                                  # noqa: BLE001 — an extra field must not cost the line
            out["stress"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, cpu, E, allcores=args.cpu_allcores)
        out["ba"]["cpu_ms"] = out["cpu_baseline"]["ba_ms"]
        out["ba"]["speedup"] = round(out["cpu_baseline"]["ba_ms"] / (t_ba_gpu * 1e3), 1)
    return out



def ba_kernel_report(run_ba):
    """north_star / SURVEY 8d: what the fastba report carries besides the time — launches per BA call, and per kernel its average duration in
    this run (torch.profiler over twelve calls behind five warm-up calls, medians), LDS bytes per workgroup and resident waves per SIMD (compiler figures,
    profiles/ba_kernel_resources.json, written by tools/kernel_resources.py --json)."""
    rep = {}
    try:
        from torch.profiler import profile, ProfilerActivity
        for _ in range(5):                                       # (the calls behind a synchronisation run on a chip that has clocked down: 31 us where
            run_ba()                                             #  rocprofv3's trace of 300 calls says 19 — warm up, take medians)
        calls = 12
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(calls):
                run_ba()
            torch.cuda.synchronize()
        ker = {}
        for ev in prof.events():
            if getattr(ev, "device_type", None) is not None and "cuda" in str(ev.device_type).lower() and ("devo::" in ev.name or "k_ba" in ev.name):
                k = ev.name.split("(")[0].replace("void ", "").replace("devo::", "")
                k = k.split("<")[0]
                ker.setdefault(k, []).append(float(getattr(ev, "device_time", 0.0) or getattr(ev, "cuda_time", 0.0)))
        if ker:
            med = {k: sorted(v)[len(v) // 2] for k, v in ker.items()}
            rep["launches"] = int(round(sum(len(v) for v in ker.values()) / calls))
            rep["kernels"] = {k: {"per_call": round(len(v) / calls, 2), "avg_us": round(med[k], 2)} for k, v in sorted(ker.items(), key=lambda kv: -sum(kv[1]))}
            rep["solve_us"] = round(sum(med[k] * len(v) for k, v in ker.items() if "solve" in k) / calls, 2)
            rep["accumulate_us"] = round(sum(med[k] * len(v) for k, v in ker.items() if "accumulate" in k or "reduce" in k) / calls, 2)
    except Exception as ex:                                      # noqa: BLE001 — the report is an extra
        rep["profiler_error"] = f"{type(ex).__name__}: {ex}"[:200]
    try:
        with open(os.path.join(ROOT, "profiles", "ba_kernel_resources.json")) as f:
            res = json.load(f)
        for k, v in rep.get("kernels", {}).items():
            r = res.get(k)
            if r:
                v.update({"lds_bytes_per_wg": r["lds_bytes"], "waves_per_simd": r["waves_per_simd"], "vgpr": r["vgpr"]})
        pick = lambda name: next((res[k] for k in res if k.startswith(name)), None)
        sname = "k_ba_solve_retract" if pick("k_ba_solve_retract") else "k_ba_solve_chain"
        sol, acc = pick(sname), pick("k_ba_accumulate_reg")
        if sol:
            rep["lds_bytes_per_wg"] = {sname: sol["lds_bytes"], "k_ba_accumulate_reg": acc["lds_bytes"] if acc else None}
            rep["waves_per_simd"] = {sname: sol["waves_per_simd"], "k_ba_accumulate_reg": acc["waves_per_simd"] if acc else None}
            rep["resources_source"] = "profiles/ba_kernel_resources.json (hipcc -Rpass-analysis=kernel-resource-usage; LDS = static + the launch's dynamic bytes)"
    except (OSError, ValueError, KeyError):
        pass
    return rep


def cpu_baseline(cfg, cpu, E, allcores=False):
    """The reference's CPU-capable path, restated (oracle/pops.py == devo/ba.py + projective_ops.py; the
    reference has no CPU corr, so oracle/altcorr.py's gather formulation stands in), timed on the host cores.
    Bounded sample (~10-20 s): transform + 2 full-size ba.py-style BA steps (median of 3, after one warm-up) +
    the 2-level lookup (all edges up to 32 768, else 8192 edges scaled to E).  torch intra-op threads are capped at 16: with one thread per
    core on a 256-core host these small-tensor ops run ~100x slower (oversubscription), which would flatter the GPU."""
    from oracle import pops, altcorr as oc
    from oracle.lie import SE3
    from devo_amd import synth
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    n, R = cfg["n"], cfg["R"]
    poses, patches, intr = cpu["poses"], cpu["patches"], cpu["intr"]
    ii, jj, kk = cpu["ii"], cpu["jj"], cpu["kk"]
    bounds = [-64, -64, cfg["W"] + 64, cfg["H"] + 64]

    def ba2():
        G, P = SE3(poses.clone()), patches.clone()
        for _ in range(2):
            G, P = pops.BA(G, P, intr, target, cpu["weight"], 1e-4, ii, jj, kk, bounds, ep=10.0, fixedp=1)

    with torch.no_grad():
        coords = pops.transform(SE3(poses), patches, intr, ii, jj, kk)           # warm-up
        t0 = time.perf_counter()
        coords = pops.transform(SE3(poses), patches, intr, ii, jj, kk)
        t_tr = time.perf_counter() - t0
        target = coords[..., 1, 1, :] + cpu["delta"]
        ba2()                                                                     # warm-up
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            ba2()
            ts.append(time.perf_counter() - t0)
        t_ba = sorted(ts)[1]
        # lookup leg: a fixed random sample of 8192 edges (all of them below that), THREE timed passes, the median scaled to E — this
        # leg swung 0.6 ... 2.5 s between boxes as a single pass (gather temporaries of the torch formulation: allocator and page-fault
        # noise), which moved `value` by 7x while the BA leg stayed within 10 %
        ns = min(E, 8192)                                                         # (in chunks of 512: the gather formulation's temporaries)
        sel = torch.randperm(E, generator=torch.Generator().manual_seed(0))[:ns]
        c2 = coords.permute(0, 1, 4, 2, 3).contiguous()[:, sel]
        f1l = synth.pyramid_l1(cpu["fmap"])

        def lookup_pass():
            for c0 in range(0, ns, 512):
                s_ = sel[c0:c0 + 512]
                oc.corr_forward(cpu["gmap"], cpu["fmap"], c2[:, c0:c0 + 512], kk[s_], jj[s_], R, acc=torch.float32)
                oc.corr_forward(cpu["gmap"], f1l, c2[:, c0:c0 + 512] / 4, kk[s_], jj[s_], R, acc=torch.float32)
        lookup_pass()                                                             # warm-up (allocator, page faults)
        tc = []
        for _ in range(3):
            t0 = time.perf_counter()
            lookup_pass()
            tc.append((time.perf_counter() - t0) * (E / ns))
        t_corr = sorted(tc)[1]
        # SURVEY 8d also asks for the single-thread figure of the BA
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        ba2()
        t_ba1 = time.perf_counter() - t0
        # ... and for every host core (os.cpu_count() threads), measured once so that the 16-thread cap above is evidence, not assertion
        t_ba_all = None
        if allcores and (os.cpu_count() or 1) > threads:
            # one thread per host core: these small-tensor ops run ~1000x slower (oversubscription: 51 s for the two-iteration solve with
            # 256 threads against 45 ms with 16) — ONE Gauss-Newton iteration, doubled, keeps the default run bounded
            torch.set_num_threads(os.cpu_count())
            G1, P1 = SE3(poses.clone()), patches.clone()
            t0 = time.perf_counter()
            pops.BA(G1, P1, intr, target, cpu["weight"], 1e-4, ii, jj, kk, bounds, ep=10.0, fixedp=1)
            t_ba_all = 2.0 * (time.perf_counter() - t0)
        torch.set_num_threads(threads)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    step_s = t_tr + t_corr + t_ba
    return {"value": round(1.0 / step_s, 4), "unit": "it/s", "cores": threads, "kind": "port", "cpu_model": model,
            "host_cores": os.cpu_count(), "ba_ms_1thread": round(t_ba1 * 1e3, 2),
            "ba_ms_allcores": (round(t_ba_all * 1e3, 2) if t_ba_all is not None else None),
            "ba_ms_allcores_note": "one torch thread per host core (ONE Gauss-Newton iteration timed, doubled): oversubscription — these small-tensor ops want <= 16 threads",
            "sample": f"torch-CPU fp32, torch.set_num_threads({threads}): transform (full, {E} edges) + 2x ba.py-style BA (full, median of 3) + "
                      f"2-level lookup on {ns} of {E} edges (median of 3 passes" + (", scaled to E)" if ns < E else ")"),
            "ba_ms": round(t_ba * 1e3, 2), "ba_ms_samples": [round(t * 1e3, 2) for t in ts], "corr_ms_scaled": round(t_corr * 1e3, 1),
            "corr_ms_samples": [round(t * 1e3, 1) for t in tc], "transform_ms": round(t_tr * 1e3, 2),
            "headline": "ba.speedup (the ba.py-style solve on the host cores against cuda_ba on the GPU: north_star's >= 10x figure); `value` adds the "
                        "lookup leg, for which the reference has no CPU implementation at all (a torch gather formulation stands in)"}


if __name__ == "__main__":
    main()
