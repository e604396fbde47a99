#!/usr/bin/env python3
"""Headline benchmark: train iters/s (+ rendered Mpix/s) on BASELINE.json config #2 —
synthetic "lego-like" scene, 300k Gaussians, 800x800, SH degree 3, one view per rank per step.

A step = one full training iteration of the hot path exactly as FusionSense drives it
(dn_splatter/dn_model.py:469-671 + trainer order): RGB+ED rasterization,
per-Gaussian normals + legacy normal rasterization, losses, backward through both rasterizers /
SH / projection, gradient all-reduce (N>1), Adam step on the 6 live parameter groups, densify
statistics.  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

_T0 = time.time()


def log(msg):
    if os.environ.get("FSGS_BENCH_VERBOSE"):
        print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--n-gauss", type=int, default=300_000)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused-caller", action="store_true",
                    help="drive the rasterizer through the drop-in rasterization()/rasterize_gaussians() "
                         "surface with the reference's op-by-op caller glue instead of the fused get_outputs node")
    ap.add_argument("--cpu-crop", type=int, default=400, help="CPU-baseline sample: central crop edge")
    ap.add_argument("--cpu-timeout", type=float, default=150.0)
    ap.add_argument("--cpu-threads", type=int, default=8)
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def cpu_baseline_worker(n_gauss: int, res: int, crop: int, threads: int):
    """Child-process body: oracle (pure-PyTorch CPU rasterizer) forward+backward on a central
    crop of view 0 of the same scene; scaled to full-frame-equivalent iters/s by pixel count."""
    from fusionsense_amd import scenes
    from fusionsense_amd.scenes import Camera
    from oracle.fusion_ref import render_fusionsense as render_ref

    torch.set_num_threads(threads)
    params = scenes.lego_like_scene(n_gauss, seed=0)
    cam = scenes.hemisphere_cameras(1, width=res, height=res, focal=1111.11 * res / 800.0, seed=0)[0]
    x0 = (cam.width - crop) // 2
    y0 = (cam.height - crop) // 2
    ccam = Camera(cam.c2w, cam.fx, cam.fy, cam.cx - x0, cam.cy - y0, crop, crop)
    cp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    t0 = time.time()
    out = render_ref(cp, ccam, sh_degree=3)
    loss = out["rgb"].mean() + 0.2 * out["depth"].mean() + 0.1 * out["normal"].mean()
    loss.backward()
    dt = time.time() - t0
    frac = (crop * crop) / float(cam.width * cam.height)
    print(json.dumps({
        "value": round(frac / dt, 6),
        "unit": "iters/s (full-frame equivalent)",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle fwd+bwd of view 0, central {crop}x{crop} crop of {cam.width}x{cam.height} "
                  f"({frac:.4f} of the pixels) in {dt:.1f}s on {threads} threads, scaled by pixel count",
    }))


def cpu_baseline(args):
    """Run the oracle in a child process (never touches the GPU) under a hard wall-clock limit."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 8))
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--n-gauss", str(args.n_gauss),
           "--res", str(args.res), "--cpu-crop", str(args.cpu_crop), "--cpu-threads", str(threads)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"value": None, "unit": "iters/s", "cores": threads, "kind": "port",
                "sample": f"oracle child failed rc={r.returncode}: {r.stderr[-200:]}"}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "iters/s", "cores": threads, "kind": "port",
                "sample": f"oracle child exceeded {args.cpu_timeout}s on a {args.cpu_crop}^2 crop"}


def pmc_record(kernel_key):
    """Counters of a kernel from the committed PMC run (profiles/pmc_traffic.json), or {}."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel_key, {}) or {}
    except (OSError, ValueError):
        return {}


def pmc_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the committed PMC run, or None."""
    return pmc_record(kernel_key).get("hbm_bytes_per_launch")


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.n_gauss, args.res, args.cpu_crop, args.cpu_threads)
        return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus or world == 1, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # FSGS_DIST_BACKEND=gloo: functional check of the multi-rank path on a box with fewer GPUs than ranks (the ranks
    # then share devices and the collectives are staged through the host); the default is RCCL, one GPU per rank
    backend = os.environ.get("FSGS_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from fusionsense_amd import frame_cache, ops, scenes
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import SplatTrainer

    W = H = args.res
    focal = 1111.11 * args.res / 800.0
    cams = scenes.hemisphere_cameras(args.views, width=W, height=H, focal=focal, seed=0)
    log('building scene')
    params = scenes.lego_like_scene(args.n_gauss, seed=0)
    log('scene built')
    # statistics only (no refinement inside the timed window: refine_every > steps)
    strategy = DensifyStrategy(SplatfactoConfig(), num_train_data=args.views, stats_only=True)
    trainer = SplatTrainer(params, dev, sh_degree=3, strategy=strategy, seed=0, fused=not args.unfused_caller)

    # targets: renders of a differently seeded scene, so gradients are non-trivial
    tgt_tr = SplatTrainer(scenes.lego_like_scene(args.n_gauss, seed=1), dev, sh_degree=3, fused=not args.unfused_caller)
    targets = []
    with torch.no_grad():
        for ci, cam in enumerate(cams):
            o = tgt_tr.forward(cam)
            torch.cuda.synchronize()
            log(f'target {ci} rendered, M={o["info"]["flatten_ids"].numel()}')
            targets.append({k: o[k].detach().clone() for k in ("rgb", "depth", "normal")})
    del tgt_tr
    torch.cuda.empty_cache()

    def view_of(step):
        return (step * world + rank) % len(cams)

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- setup: size the workspace pool for every view (forward + backward, no optimizer step), so that
    # neither the warmup nor the timed steps meet a first-time device allocation ----
    if not os.environ.get("FSGS_BENCH_NO_PRIME"):
        for v in range(len(cams)):
            trainer.train_step(cams[v], targets[v], optimizer_step=False)
        torch.cuda.synchronize()
        trainer.step = 0
        log('workspace primed')

    # ---- warmup ----
    for s in range(args.warmup):
        v = view_of(s)
        trainer.train_step(cams[v], targets[v])
        torch.cuda.synchronize()
        log(f'warmup step {s} done')
    torch.cuda.synchronize()
    barrier()

    # ---- timed region: EXACTLY args.steps full iterations ----
    # only the dominant kernel is event-timed inside the timed region (pre-created events): timing a launch costs
    # host time right in front of it, i.e. a GPU bubble; every other kernel is timed in the untimed pass below
    ops.TIMER.reset(enabled=not os.environ.get("FSGS_BENCH_NO_TIMER"), only=("raster_bwd",), prealloc=args.steps + 2)
    # keep the interpreter's cyclic collector out of the timed region (a generation-2 pass costs tens of ms).
    # No gc.collect() here: freeing the setup's garbage right now reshuffles the caching allocator's pools and
    # was measured to cost 7 % in the steps that follow.
    gcm = os.environ.get("FSGS_BENCH_GC", "fd")
    if "c" in gcm:
        gc.collect()
    if "f" in gcm:
        gc.freeze()
    if "d" in gcm:
        gc.disable()
    torch.cuda.synchronize()
    n_alloc0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
    t0 = time.perf_counter()
    step_ends = []
    for s in range(args.steps):
        v = view_of(args.warmup + s)
        trainer.train_step(cams[v], targets[v])
        step_ends.append(time.perf_counter())  # (every step waits for its live-pair count: host time tracks the GPU)
    trainer.flush()  # (data-parallel runs: the last step's deferred feature update belongs to the timed work)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    gc.unfreeze()
    kernel_ms = ops.TIMER.summary()
    if world > 1 and os.environ.get("FSGS_BENCH_CHECK_REPLICAS"):
        # the ranks must hold bit-identical parameters after the timed steps (rank-ordered gradient sums)
        for name, prm in trainer.params.items():
            ref = prm.data.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(ref, prm.data), f"rank {rank}: replica of {name} differs from rank 0"
        log('replicas identical')
    n_alloc = torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - n_alloc0
    log(f'timed region done: {elapsed:.3f}s')
    ops.TIMER.reset(enabled=False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- side measurements (outside the timed region) ----
    ops.TIMER.reset(enabled=True)
    for s in range(min(args.steps, 10)):
        v = view_of(s)
        trainer.train_step(cams[v], targets[v])
    torch.cuda.synchronize()
    kernel_ms_all = ops.TIMER.summary()
    ops.TIMER.reset(enabled=False)
    for k_, v_ in kernel_ms_all.items():
        kernel_ms.setdefault(k_, v_)  # heavy kernels keep their in-region figures
    info = trainer.last_info
    N = trainer.num_gaussians()
    # M = gsplat's intersection count (sum of tiles_per_gauss): the unit of the byte model; the fused
    # node bins and sorts only the live subset
    M = int(info["tiles_per_gauss"].sum().item())
    n_live = int(info["flatten_ids"].numel())
    n_vis = int((info["radii"] > 0).sum().item())
    P = W * H
    # iteration without the optimizer step
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for s in range(args.steps):
        v = view_of(s)
        trainer.train_step(cams[v], targets[v], optimizer_step=False)
    torch.cuda.synchronize()
    t_noopt = (time.perf_counter() - t1) / args.steps
    # forward-only latency -> rendered Mpix/s (num_rays_per_sec of dn_pipeline.py:246-248)
    with torch.no_grad():
        for s in range(2):
            trainer.forward(cams[view_of(s)])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for s in range(args.steps):
            trainer.forward(cams[view_of(s)])
        torch.cuda.synchronize()
        t_fwd = (time.perf_counter() - t2) / args.steps

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        iters_per_s = world * args.steps / elapsed
        # algorithmic bytes.  The normal plane reuses the RGB+ED pass's sorted
        # lists when both bbox rules bin the frame identically (+80 B/isect), else re-bins (+244).
        key_bits = 32 + ops.tile_bits((W // 16 + (W % 16 > 0)) * (H // 16 + (H % 16 > 0)))
        sort_b = 2 * 12 * ((key_bits + 7) // 8)
        b_isect = 12 + sort_b + 8 + 44 + 44
        fused = not args.unfused_caller
        reused = fused or (frame_cache.hits > 0 and frame_cache.misses == 0)
        b_isect_normal = (40 + 40) if reused else (12 + sort_b + 8 + 40 + 40)
        b_iter = N * 352 + n_vis * 444 + M * (b_isect + b_isect_normal) + P * 92
        # dominant kernel = the libfsgs launch with the largest share of the step; algorithmic bytes
        # per launch from the per-unit figures of SURVEY.md §8d
        alg = {
            "raster_bwd_quad_d4e3": ("raster_bwd_live_kernel<4,true,3> (RGB+ED and normal plane, one walk)",
                                     M * (44 + 40) + P * (28 + 20) + n_vis * (48 + 28)),
            "raster_fwd_quad_d4e3": ("raster_fwd_quad_kernel<4,3> (filter + gather + composite, RGB+ED and normal plane)",
                                     M * (44 + 40) + P * (24 + 20)),
            "raster_bwd_quad_d4": ("raster_bwd_live_kernel<4,true,0>", M * 44 + P * 28 + n_vis * 48),
            "raster_bwd_quad_d3": ("raster_bwd_live_kernel<3,false,0>", M * 40 + P * 20 + n_vis * 28),
            "raster_fwd_quad_d4": ("raster_fwd_quad_kernel<4,0>", M * 44 + P * 24),
            "raster_fwd_quad_d3": ("raster_fwd_quad_kernel<3,0>", M * 40 + P * 20),
            "raster_bwd_d4": ("raster_bwd_kernel<4,true>", M * 44 + P * 28 + n_vis * 48),
            "sort_pairs": ("radix sort (hist + scan + scatter per 8-bit pass)", M * sort_b),
            "tile_sort": ("partition by tile + per-tile LDS sort", M * sort_b),
        }
        cand = [(v["avg_ms"], k) for k, v in kernel_ms.items() if k in alg]  # (each is launched once per step)
        roofline = None
        if cand:
            _, dom = max(cand)
            dom_ms = kernel_ms[dom]["avg_ms"]
            name, dom_bytes = alg[dom]
            ach = dom_bytes / (dom_ms * 1e-3) / 1e9
            roofline = {"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                        # PMC counters cannot be read from inside the timed process: `traffic` is the
                        # rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE figure (bytes per launch, gfx950-corrected)
                        # of this kernel from the committed run in profiles/ (DESIGN.md §5), null if absent
                        "traffic": pmc_traffic(dom), "algorithmic_bytes": dom_bytes,
                        "avg_launch_ms": round(dom_ms, 4)}
            # the compositing kernels are bound by fp32 VALU issue, not by HBM (the contract's `bound` has no
            # such value): say how busy the vector ALUs are, from the committed SQ counters and THIS run's time
            q = pmc_record(dom).get("sq_active_inst_valu_quadcycles_per_launch")
            if q:
                roofline["valu_busy_frac"] = round(4.0 * q / (1024 * dom_ms * 1e-3 * 2.4e9), 4)
                roofline["limiter"] = "fp32 VALU issue (see DESIGN.md 5)"
        line = {
            "metric": "train_iters_per_s",
            "value": round(iters_per_s, 3),
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE config #2: synthetic lego-like, {N} Gaussians, {W}x{H}, "
                                   "SH deg 3, RGB+ED + normal pass, fwd+bwd+Adam+densify stats, "
                                   "1 view/rank/step; caller = " + ("fused get_outputs node" if fused else
                                                                    "reference-style op-by-op glue over the drop-in surface"),
                       "n_gaussians": N, "width": W, "height": H, "views": len(cams),
                       "n_isects": M, "n_isects_live": n_live, "n_visible": n_vis, "isects_per_gaussian": round(M / max(N, 1), 3),
                       "normal_pass_list_reuse": bool(reused), "parallelism": f"dp{world}"},
            "rendered_mpix_per_s": round(world * P / t_fwd / 1e6, 2),
            "fwd_ms": round(t_fwd * 1e3, 3),
            "iters_per_s_excl_optimizer": round(world / t_noopt, 3),
            "iter_algorithmic_bytes": b_iter,
            "iter_hbm_frac": round(b_iter / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9), 5),
            "device_mallocs_in_timed_region": n_alloc,
            "max_step_ms": round(1e3 * max(b - a for a, b in zip([t0] + step_ends[:-1], step_ends)), 3),
            "slowest_step": max(range(len(step_ends)), key=lambda i: step_ends[i] - ([t0] + step_ends)[i]),
            "kernels_ms": kernel_ms,
            "roofline": roofline,
        }
        log('gpu part done')
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
