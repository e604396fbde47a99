#!/usr/bin/env python3
"""Headline benchmark: train iters/s (+ rendered Mpix/s) of the Gaussian-splatting hot path.

Default (what the driver runs): BASELINE.json config #2 — synthetic "lego-like" scene, 300k Gaussians, 800x800,
SH degree 3, 100 hemisphere cameras, one view per rank per step.  ``--config 3|4|5`` run the other single-GPU-sized
BASELINE configurations through the same loop (they are parity-test cases first; their lines are for DESIGN.md):
  3  FusionSense-shaped scene: 1280x720, 9 views, ~60k seeds, RGB + sensor depth + mono normals + mask + 5 touch
     patches, the reference's get_loss_dict, a 300-step window of the real schedule (steps 950..1249: touch patches
     added at 1000, three refinements, binary opacities, SH degree 0 -> 1)
  4  "bicycle-like": 6 M Gaussians, 1920x1080, 8 views (one view per rank per step when launched on 8 GPUs)
  5  10 M Gaussians with half-precision attribute storage (fp32 master + Adam), 1920x1080

A step = one full training iteration of the hot path exactly as FusionSense drives it
(dn_splatter/dn_model.py:469-671 + trainer order): RGB+ED rasterization, per-Gaussian normals +
normal rasterization, losses, backward through both rasterizers / SH / projection, gradient exchange (N>1), Adam
step on the 6 live parameter groups, densify statistics (config 3: + the callbacks).  Prints ONE JSON line (rank 0).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime forces a host wait every DEBUG_CLR_MAX_BATCH_SIZE (1000) commands enqueued without a synchronisation
# in between — here every ~77 steps of 13 commands.  Normally that costs 0.1 ms; in ~1 run of 15 the wait went to sleep
# for 5 + 15 ms on the GPU boxes of this pool (always at timed step 77-78, whatever view was there; 8 of 119 default runs
# against 0 of 50 with the limit raised — DESIGN.md §7).  A training loop synchronises long before 16384 commands
# (its refinement callbacks read mask counts every 100 steps), so the limit is moved out of the way, before the runtime
# loads; an explicit setting of the caller's wins.
RUNTIME_ENV = {"DEBUG_CLR_MAX_BATCH_SIZE": "16384"}
for _k, _v in RUNTIME_ENV.items():
    RUNTIME_ENV[_k] = os.environ.setdefault(_k, _v)

import torch
import torch.distributed as dist

_T0 = time.time()


def gpu_step_stats(events, views, every=1):
    """GPU-side duration of the timed steps (event to event on the step's stream): quantiles, and — the views differ
    in work — the largest ratio of a step to the median of the steps of ITS view when views repeat (host stalls and
    redone frames show here; a different view does not).  ``every`` > 1: an event was recorded behind every ``every``-th
    step only (an event record is a marker packet: ~6 us of idle GPU in front of the next launch, 1.3 % of a config-#2
    step) — the figures are then per-step MEANS over groups of ``every`` consecutive steps."""
    d = [events[i].elapsed_time(events[i + 1]) / every for i in range(len(events) - 1)]
    if not d:
        return None
    q = sorted(d)
    out = {"p50": round(q[len(q) // 2], 4), "p99": round(q[min(len(q) - 1, int(0.99 * len(q)))], 4),
           "max": round(q[-1], 4)}
    if every > 1:
        out["per"] = f"mean of {every} consecutive steps"
        views = []  # (groups mix views)
    if len(d) >= 8:  # drift over the run (clocks, growing state): the mean of each quarter
        n4 = len(d) // 4
        out["mean_by_quarter"] = [round(sum(d[k * n4:(k + 1) * n4]) / n4, 4) for k in range(4)]
    by_view = {}
    for v, t in zip(views, d):
        by_view.setdefault(v, []).append(t)
    ratios = []
    for v, ts in by_view.items():
        if len(ts) >= 3:
            med = sorted(ts)[len(ts) // 2]
            ratios.append(max(ts) / med)
    if ratios:
        out["max_over_view_median"] = round(max(ratios), 4)
    return out


def log(msg):
    if os.environ.get("FSGS_BENCH_VERBOSE"):
        print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)
DEFAULTS = {2: dict(steps=100, warmup=10, views=100), 3: dict(steps=300, warmup=10, views=9),
            4: dict(steps=20, warmup=8, views=8), 5: dict(steps=10, warmup=4, views=4)}
# render tolerance of the fp32 path against the reference semantics: images 1e-4 of the tensor
# maximum, gradients 3e-3 of each tensor's own maximum (fp32 atomics reorder sums); index outputs bit-exact
RENDER_TOLERANCE = {"forward_rel": 1e-4, "gradient_rel": 3e-3, "indices": "bit-exact"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5])
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--n-gauss", type=int, default=None)
    ap.add_argument("--res", type=int, default=800, help="config 2: square image edge")
    ap.add_argument("--views", type=int, default=None)
    ap.add_argument("--start-step", type=int, default=950,
                    help="config 3: first step of the timed window of the 15 000-step schedule (0 + --steps 15000 = all of it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the side measurement through the drop-in surface")
    ap.add_argument("--unfused-caller", action="store_true",
                    help="drive the rasterizer through the drop-in rasterization()/rasterize_gaussians() "
                         "surface with the reference's op-by-op caller glue instead of the fused get_outputs node")
    ap.add_argument("--cpu-crop", type=int, default=280, help="CPU-baseline sample: central crop edge")
    ap.add_argument("--cpu-views", type=int, default=3)
    ap.add_argument("--cpu-timeout", type=float, default=90.0)
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline: cap on the threads (0 = all host cores)")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--spawn-timeout", type=float, default=float(os.environ.get("FSGS_SPAWN_TIMEOUT", "900")),
                    help="--gpus N without a launcher: seconds after which the ranks' process group is killed")
    ap.add_argument("--scene", default=None, choices=["surface", "volume"],
                    help="configs 4 / 5: 'surface' (default) = ground + objects seen from a ring outside the content; "
                         "'volume' = the round-1 fog of Gaussians with the cameras inside it")
    ap.add_argument("--fwd-walk", type=int, default=None, choices=[0, 1],
                    help="A/B: the forward compositing's walk (0 = four waves per quadrant, 1 = one wave); default = "
                         "the product's (fused.FWD_WALK: one wave per quadrant + hand-off of long walks)")
    ap.add_argument("--handoff", type=int, default=None,
                    help="A/B: records a wave streams before it hands the rest of its list to the tail workgroups "
                         "(0 = never; default = fused.FWD_WALK.handoff_records)")
    ap.add_argument("--handoff-rel-len", type=int, default=None, help="A/B: hand off only lists longer than this many "
                    "times the frame's mean list length")
    ap.add_argument("--handoff-gate-len", type=int, default=None, help="A/B: hand off only in frames whose longest list "
                    "exceeds this many times the mean list length (0 = any frame)")
    ap.add_argument("--fwd-order", type=int, default=None, help="A/B: 0 = the forward's two-pass long-lists-first order")
    ap.add_argument("--sort-order", type=int, default=None, help="A/B: 0 = the in-tile sorts in tile order")
    ap.add_argument("--bwd-order", type=int, default=None, help="A/B: 0 = the backward's quadrants in the size rule's order")
    ap.add_argument("--bin-fill", type=int, default=None, help="A/B: 0 = two-pass binning on every frame")
    ap.add_argument("--densify", action="store_true",
                    help="config 2: run the schedule's refinements (split / dup / cull every 100 steps from step 600 on) "
                         "inside the timed region, e.g. --steps 300: the count-free binning's hit rate over an epoch with "
                         "densification (binning_frames, live_list_overflows) instead of over frozen model sizes")
    ap.add_argument("--express", type=int, default=None, choices=[0, 1],
                    help="A/B: 0 = every step launch by launch from Python (no fsgs_step_run)")
    ap.add_argument("--bwd-queue", type=int, default=None, help="A/B: queue positions for the backward's extra workgroups")
    ap.add_argument("--tail-items", type=int, default=None, help="A/B: queue positions of the hand-off (= tail workgroups)")
    a = ap.parse_args()
    d = DEFAULTS[a.config]
    a.steps = d["steps"] if a.steps is None else a.steps
    a.warmup = d["warmup"] if a.warmup is None else a.warmup
    a.views = d["views"] if a.views is None else a.views
    if a.n_gauss is None:
        a.n_gauss = {2: 300_000, 3: 60_000, 4: 6_000_000, 5: 10_000_000}[a.config]
    return a


# ------------------------------------------------------------------------------------------------------------
# CPU baseline (child process, never touches the GPU): the oracle on the host cores
# ------------------------------------------------------------------------------------------------------------
def cpu_baseline_worker(n_gauss: int, res: int, crop: int, threads: int, n_views: int):
    """oracle (pure-PyTorch CPU rasterizer) forward+backward of config #2 on a central crop of ``n_views`` views
    (median, scaled to full-frame-equivalent iters/s by pixel count) + config #1 (1k cube, 128x128) in full."""
    from fusionsense_amd import scenes
    from fusionsense_amd.scenes import Camera
    from oracle.fusion_ref import render_fusionsense as render_ref

    torch.set_num_threads(threads)

    def fwd_bwd(params, cam):
        cp = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        t0 = time.time()
        out = render_ref(cp, cam, sh_degree=3)
        (out["rgb"].mean() + 0.2 * out["depth"].mean() + 0.1 * out["normal"].mean()).backward()
        return time.time() - t0

    p1, c1 = scenes.cube_scene(1000, seed=0)
    t_cfg1 = fwd_bwd(p1, c1)
    params = scenes.lego_like_scene(n_gauss, seed=0)
    cams = scenes.hemisphere_cameras(n_views, width=res, height=res, focal=1111.11 * res / 800.0, seed=0)
    times = []
    for cam in cams:
        x0, y0 = (cam.width - crop) // 2, (cam.height - crop) // 2
        times.append(fwd_bwd(params, Camera(cam.c2w, cam.fx, cam.fy, cam.cx - x0, cam.cy - y0, crop, crop)))
    dt = statistics.median(times)
    frac = (crop * crop) / float(res * res)
    print(json.dumps({
        "value": round(frac / dt, 6),
        "unit": "iters/s (full-frame equivalent)",
        "cores": threads,
        "kind": "port, extrapolated from crop",
        "config1_full_iters_per_s": round(1.0 / t_cfg1, 4),
        "sample": f"oracle fwd+bwd, config #2: median of {n_views} views, central {crop}x{crop} crop of {res}x{res} "
                  f"({frac:.4f} of the pixels; {', '.join(f'{t:.1f}' for t in times)} s) scaled by pixel count; "
                  f"config #1 (1k cube, 128x128) in full: {t_cfg1:.2f} s; {threads} threads",
    }))


def cpu_baseline(args):
    """The oracle on this box's host cores.  A pure-PyTorch rasterizer does not scale with threads (measured on the
    256-core host of the GPU boxes, round 4: one small view takes 2.7 s on 8 threads, 2.6 s on 16, 5.4 s on 64 and does
    not finish in 25 s on 256), so the thread count is CHOSEN by a bounded probe — one view of a 96^2 crop at 16 and at 8
    threads, 8 s each at most (round 5: the 64- and all-core probes, which only ever lost, are gone: they cost 30 s of
    every driver run) — then the full sample (``--cpu-views`` views, ``--cpu-crop``^2 crop) at the faster count.
    ``cores`` = the threads of the quoted figure, ``host_cores_available`` = what the process may run on,
    ``thread_sweep_s`` = the probe times (None = did not finish).  The figure is an EXTRAPOLATION from the crop to the
    full frame by pixel count (``extrapolated``)."""
    import subprocess
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    if args.cpu_threads and args.cpu_threads > 0:
        avail = min(avail, args.cpu_threads)

    def child(threads, views, crop, timeout):
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--n-gauss", "300000",
               "--res", "800", "--cpu-crop", str(crop), "--cpu-threads", str(threads), "--cpu-views", str(views)]
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="")
        t0 = time.time()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                return json.loads(lines[-1]), time.time() - t0
            return {"value": None, "unit": "iters/s", "cores": threads, "kind": "port",
                    "sample": f"oracle child failed rc={r.returncode}: {r.stderr[-200:]}"}, None
        except subprocess.TimeoutExpired:
            return {"value": None, "unit": "iters/s", "cores": threads, "kind": "port",
                    "sample": f"oracle child exceeded {timeout}s on {views} {crop}^2 crops with {threads} threads"}, None

    candidates = sorted({t for t in (16, 8) if t <= avail} or {avail}, reverse=True)
    sweep = {}
    best = None
    if len(candidates) > 1:
        for t in candidates:
            res, secs = child(t, 1, 96, 8.0)
            ok = res.get("value") is not None and secs is not None
            sweep[str(t)] = round(secs, 1) if ok else None
            if ok and (best is None or secs < sweep[str(best)]):
                best = t
    if best is None:
        best = candidates[-1]
    out, _ = child(best, args.cpu_views, args.cpu_crop, args.cpu_timeout)
    out["host_cores_available"] = avail
    out["thread_sweep_s"] = sweep or None
    out["extrapolated"] = (f"from a central {args.cpu_crop}x{args.cpu_crop} crop of the 800x800 frame, scaled by pixel count; "
                           "more threads only lose (round 4 on this host class: 64 threads 5.4 s, 256 > 25 s for the 2.6 s probe)")
    return out


def bwd_dispatch_choice(dev, W, H):
    try:
        import fusionsense_amd.fused as _f
        return {"stride": int(_f.BWD_DISPATCH.stride(W, H)), "rule": "7 apart up to 2^20 pixels, row-major above",
                "longest_share_first": bool(_f.BWD_ORDER and _f.BWD_QUEUE_ITEMS > 0),  # (models below 2^20 Gaussians)
                "tuning_frames": 0}
    except Exception:
        return None


def fwd_walk_choice(dev, W, H, n_gaussians=0):
    """The forward compositing's walk of this run (fixed since round 5: fused.FWD_WALK): walk 1 = one wave per quadrant,
    ``handoff_records`` > 0 = long walks finished chunk-parallel by the launch's tail workgroups; ``tail_error`` = 1 if a
    look-back wait of that pass ever ran into its bound (results invalid; never observed)."""
    try:
        from fusionsense_amd.fused import FWD_WALK
        walk, handoff = FWD_WALK.choice(n_gaussians)
        return {"walk": int(walk), "handoff_records": int(handoff), "forced": bool(FWD_WALK.forced),
                "handoff_rel_len": int(FWD_WALK.handoff_rel_len), "handoff_gate_len": int(FWD_WALK.handoff_gate_len),
                "tail_workgroups_max": int(FWD_WALK.max_items),
                # tail workgroups the launches carry NOW (0 while no list of this run's frames asked for a hand-off
                # lately: performance only, the results do not depend on it) and hand-offs asked for so far
                "tail_workgroups_now": FWD_WALK.tail_items_now(dev) if handoff else 0,
                "handoffs_seen": int((FWD_WALK.state.get(str(dev)) or {}).get("seen", 0)) if handoff else 0,
                "tail_error": FWD_WALK.error(dev) if handoff else 0,
                "tuning_frames": 0}
    except Exception:
        return None


def never_updated_frac(trainer):
    try:
        p = trainer._params["opacities"]
        v = trainer.optimizers["opacities"].state[p]["exp_avg_sq"]
        return round(float((v.reshape(-1) == 0).float().mean()), 4)
    except Exception:  # (no optimizer state yet)
        return None


def trainer_last_alpha(trainer):
    """The accumulation image of the trainer's last frame (kept by the side-measurement loop below)."""
    return trainer._bench_last_out["accumulation"]


def pmc_record(kernel_key, config):
    """Counters of a kernel from the committed PMC run of this round (profiles/pmc_traffic.json), or {}.
    PMC counters cannot be read from inside the timed process; tools/pmc_summary.py writes this file from the
    rocprofv3 --pmc passes of the same bench command."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        return d.get(f"config{config}", {}).get(kernel_key, {}) or {}
    except (OSError, ValueError):
        return {}


# kernel key of bench.py -> entry of profiles/r6_isa_mix.json (tools/isa_mix_all.sh: the hot loop's static instruction mix)
_ISA_MIX_KEY = {"raster_bwd_quad_d4e3": "raster_bwd_live<4,true,3>", "raster_fwd_quad_d4e3": "raster_fwd_wave<4,3>",
                "gaussian_bwd": "gauss_sh_bwd", "ssim_l1_fwd": "ssim_l1_fwd", "ssim_l1_bwd": "ssim_l1_bwd",
                "isect_count_live": "isect_live_bin_fill", "tile_sort": "tile_sort_kernel2"}


def isa_mix_record(kernel_key):
    """{"simple_share", "dpp_share", "trans_share", "loop"} of a kernel's hot loop, or {} (then every instruction is priced
    as a full 4-cycle pass)."""
    path = os.path.join(ROOT, "profiles", "r6_isa_mix.json")
    try:
        with open(path) as f:
            return json.load(f).get(_ISA_MIX_KEY.get(kernel_key, kernel_key), {}) or {}
    except (OSError, ValueError):
        return {}


# ------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------
def render_targets(tr, cams, keys=("rgb", "depth", "normal", "accumulation")):
    out = []
    with torch.no_grad():
        for cam in cams:
            o = tr.forward(cam)
            out.append({k: o[k].detach().clone() for k in keys})
    torch.cuda.synchronize()
    return out


def build_workload(args, dev):
    from fusionsense_amd import scenes
    from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
    from fusionsense_amd.trainer import SplatTrainer
    fused = not args.unfused_caller
    w = dict(start_step=0, extra={})
    if args.config == 2:
        W = H = args.res
        cams = scenes.hemisphere_cameras(args.views, width=W, height=H, focal=1111.11 * args.res / 800.0, seed=0)
        params = scenes.lego_like_scene(args.n_gauss, seed=0)
        # (stats_only: the statistics of after_train are kept every step, the refinement itself is not run — unless
        # --densify asks for the real schedule; the scene then starts 600 steps in, behind the warm-up of dn_model.py:116)
        strategy = DensifyStrategy(SplatfactoConfig(), num_train_data=args.views, stats_only=not args.densify)
        trainer = SplatTrainer(params, dev, sh_degree=3, strategy=strategy, seed=0, fused=fused)
        if args.densify:
            w.update(start_step=600)
        tgt = SplatTrainer(scenes.lego_like_scene(args.n_gauss, seed=1), dev, sh_degree=3, fused=fused)
        targets = [{k: t[k] for k in ("rgb", "depth", "normal")} for t in render_targets(tgt, cams)]
        w.update(name=f"BASELINE config #2: synthetic lego-like, {args.n_gauss} Gaussians, {W}x{H}, SH deg 3, "
                      f"{args.views} hemisphere cameras, RGB+ED + normal pass, benchmark loss (L1+SSIM rgb, L1 depth, "
                      "L1 normal), fwd+bwd+Adam+densify stats", params=params)
    elif args.config == 3:
        from fusionsense_amd.losses import LossConfig, prepare_batch
        n_hull = args.n_gauss // 3
        params, cams = scenes.fusionsense_like_scene(seed=0, n_hull=n_hull, n_bg=args.n_gauss - n_hull)
        g = torch.Generator().manual_seed(5)
        params["opacities"] = 1.5 * torch.randn(params["opacities"].shape, generator=g)  # ~950 steps into training
        cams = cams[:args.views]
        H, W = cams[0].height, cams[0].width
        strategy = DensifyStrategy(SplatfactoConfig(), num_train_data=len(cams))
        patches = scenes.touch_patches_on_blob(5, 2000, radius=0.1, seed=0)
        strategy.set_metadata(touch_patches=patches, gel_scale_factor=6.34e-5, add_touch_at=1000,
                              visual_hull=params["means"][:n_hull].clone(), scale_factor=1.0)
        cfg = LossConfig()
        trainer = SplatTrainer(params, dev, sh_degree=3, strategy=strategy, seed=0, fused=fused,
                               sh_degree_interval=1000, loss_cfg=cfg)
        tp, _ = scenes.fusionsense_like_scene(seed=1, n_hull=n_hull, n_bg=args.n_gauss - n_hull)
        tp["opacities"] = tp["opacities"] + 3.0
        tgt = SplatTrainer(tp, dev, sh_degree=3, fused=fused)
        yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
        mask = (((xx - W / 2) / (0.42 * W)) ** 2 + ((yy - H / 2) / (0.46 * H)) ** 2 < 1.0)[..., None]
        targets = []
        for t in render_targets(tgt, cams):
            sensor = torch.where(t["accumulation"] > 0.5, t["depth"], torch.zeros_like(t["depth"]))  # holes
            targets.append(prepare_batch({"image": t["rgb"], "sensor_depth": sensor, "normal": t["normal"],
                                          "mask": mask}, cfg, dev))
        # One-time costs of the callbacks' torch operators (hipBLAS handle for the 3x3 box transform, lazily loaded
        # code objects: ~1.2 s in a fresh process, measured) belong to a 15 000-step run's first refinement, not to
        # a 300-step window: run every callback once on a small throwaway trainer.
        pp, _ = scenes.fusionsense_like_scene(seed=2, n_hull=2000, n_bg=4000)
        ps = DensifyStrategy(SplatfactoConfig(), num_train_data=len(cams))
        ps.set_metadata(touch_patches=patches, add_touch_at=1000, visual_hull=pp["means"][:2000].clone())
        ptr_ = SplatTrainer(pp, dev, sh_degree=3, strategy=ps, seed=0, fused=fused, sh_degree_interval=1000, loss_cfg=cfg)
        ptr_.step = 999
        for _ in range(102):
            ptr_.train_step(cams[0], targets[0])
        del ptr_, ps, pp
        w.update(name=f"BASELINE config #3: FusionSense-shaped scene, {args.n_gauss} seed Gaussians (1/3 object blob), "
                      f"{W}x{H}, {len(cams)} views, RGB + sensor depth + mono normals + mask + 5 touch patches x ~2000 "
                      f"points; loss = the reference's get_loss_dict; steps {args.start_step}..: add_touch_patch at 1000, refinement "
                      "(densify / cull / hull + touch pruning) every 100, binary opacities, SH degree step//1000",
                 params=params, start_step=args.start_step)
    else:
        W, H = 1920, 1080
        scene = args.scene or "surface"
        if scene == "surface":
            # round 4: Gaussians on surfaces (ground + boxes), cameras on a ring OUTSIDE the content — every surface faces
            # some camera, pixels saturate on the first surfaces, what lies behind them is occluded (not empty)
            kw = dict(kv.split("=") for kv in os.environ.get("FSGS_SCENE_PARAMS", "").split(",") if "=" in kv)
            kw = {k: (int(v) if k == "n_objects" else float(v)) for k, v in kw.items()}
            ring = {k: kw.pop(k) for k in ("radius", "height") if k in kw}
            cams = scenes.ring_cameras(args.views, width=W, height_px=H, focal=1500.0, **ring)
            make = lambda seed: scenes.surface_dense_scene(args.n_gauss, seed=seed, **kw)  # noqa: E731
            what = "surface-bearing (ground disc + 1200 boxes, discs lying in their surfaces, bimodal opacities, ring cameras outside)"
        else:
            cams = []
            for i in range(args.views):
                az = 2 * math.pi * i / args.views
                eye = torch.tensor([2.4 * math.cos(az), 2.4 * math.sin(az), 0.9])
                cams.append(scenes.Camera(scenes.look_at_c2w(eye, torch.tensor([0.0, 0.0, 0.0])), 1500.0, 1500.0, W / 2.0,
                                          H / 2.0, W, H))
            make = lambda seed: scenes.bicycle_like_scene(args.n_gauss, seed=seed)  # noqa: E731
            what = "volume (round-1 recipe: a fog of small Gaussians with the cameras inside it)"
        params = make(0)
        strategy = DensifyStrategy(SplatfactoConfig(), num_train_data=args.views, stats_only=True)
        half = args.config == 5
        trainer = SplatTrainer(params, dev, sh_degree=3, strategy=strategy, seed=0, fused=fused,
                               **({"half_attributes": True} if half else {}))
        tgt = SplatTrainer(make(1), dev, sh_degree=3, fused=fused)
        targets = [{k: t[k] for k in ("rgb", "depth", "normal")} for t in render_targets(tgt, cams)]
        name = ("BASELINE config #4: synthetic bicycle-like" if not half else
                "BASELINE config #5: fp16 attribute storage (features, scales, quats, opacities; fp32 means, fp32 master + "
                "Adam), synthetic bicycle-like")
        w.update(name=f"{name}, scene = {what}, {args.n_gauss} Gaussians, {W}x{H}, SH deg 3, {args.views} views, "
                      "benchmark loss, fwd+bwd+Adam+densify stats", params=params, scene=scene)
    try:
        del tgt
    except NameError:
        pass
    torch.cuda.empty_cache()
    w.update(trainer=trainer, cams=cams, targets=targets, W=cams[0].width, H=cams[0].height)
    return w


def visible_gpu_count() -> int:
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (the parent of the ranks must never
    initialise the GPU): the KFD topology's nodes with SIMDs, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES when set.  Falls back to torch.cuda.device_count() (which does not create a context on this
    image) only where the topology is unreadable."""
    n = None
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        n = None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = len([x for x in v.split(",") if x.strip() != ""])
            n = listed if n is None else min(n, listed)
    if n is None:
        n = torch.cuda.device_count()
    return n


def spawn_ranks(args) -> int:
    """``--gpus N`` (N > 1) WITHOUT a launcher: start the N ranks as a CHILD ``python -m torch.distributed.run`` (one
    process per GPU, 127.0.0.1 rendezvous on a free port) in a session of its own, BEFORE this process has made any GPU
    call, relay rank 0's JSON line and return the child's exit code.  Nothing is re-executed in place (a process that has
    initialised the GPU must never exec), and a failed attempt is a non-zero exit, never a silent one-rank run (the
    reference's hook for the multi-GPU path: /root/reference/dn_splatter/dn_pipeline.py:162-167).
    Watchdog: after ``--spawn-timeout`` seconds (default 900, well under the driver's limit) the child's whole process
    group is killed, the tail of what the ranks printed is shown and the exit code is 124 — a hung rendezvous or
    collective must not hold the GPUs until someone else's clock runs out."""
    import collections
    import signal
    import socket
    import subprocess
    import threading
    n_dev = visible_gpu_count()
    backend = os.environ.get("FSGS_DIST_BACKEND", "nccl")
    if backend == "nccl" and n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} over RCCL needs {args.gpus} visible GPUs, found {n_dev} "
              "(FSGS_DIST_BACKEND=gloo runs the ranks on shared devices: a functional check, not a measurement)",
              file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, FSGS_BENCH_SPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT,
                         start_new_session=True)  # (its own process group: the ranks can be reaped together)
    lines, tail = [], collections.deque(maxlen=60)

    def pump():
        for ln in p.stdout:
            if ln.startswith("{"):
                lines.append(ln.rstrip("\n"))
            else:
                tail.append(ln)
                sys.stderr.write(ln)
    t = threading.Thread(target=pump, daemon=True)
    t.start()
    try:
        rc = p.wait(timeout=args.spawn_timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGTERM)
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait(timeout=10)
        except (ProcessLookupError, subprocess.TimeoutExpired):
            pass
        t.join(timeout=5)
        print(f"bench.py: the {args.gpus} ranks did not finish within --spawn-timeout {args.spawn_timeout:.0f} s: their "
              f"process group was killed.  Last output of the ranks:\n{''.join(tail)}", file=sys.stderr)
        return 124
    t.join(timeout=10)
    if rc == 0 and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 3
    if rc == 0:
        print(lines[0], flush=True)
    elif tail:
        print(f"bench.py: the ranks exited with code {rc}", file=sys.stderr)
    return rc


def ranks_seen(rank, local_dev, backend, dev):
    """All-gather of (rank, device index): what the group really consists of, for the bench line."""
    world = dist.get_world_size()
    mine = torch.tensor([rank, local_dev], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    got = [g.cpu().tolist() for g in got]
    return [g[0] for g in got], [g[1] for g in got]


def main():
    args = parse()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.n_gauss, args.res, args.cpu_crop, args.cpu_threads, args.cpu_views)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE {world} != --gpus {args.gpus}: refusing to measure a different job than the one "
              "asked for", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("FSGS_BENCH_TEST_HANG") == "1":  # (test hook: a rank that never reaches the rendezvous)
        print(f"rank {rank}: hanging for the watchdog test", file=sys.stderr, flush=True)
        time.sleep(3600)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # FSGS_DIST_BACKEND=gloo: functional check of the multi-rank path on a box with fewer GPUs than ranks (the ranks
    # then share devices and the collectives are staged through the host); the default is RCCL, one GPU per rank
    backend = os.environ.get("FSGS_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    elif world > 1 and torch.cuda.device_count() < world:
        print(f"bench.py: {world} RCCL ranks need {world} GPUs, found {torch.cuda.device_count()}", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # (FSGS_FORCE_COLLECTIVES=1 under a one-rank launcher: the multi-GPU step's RCCL calls on a one-GPU box)
    one_rank_group = world == 1 and os.environ.get("FSGS_FORCE_COLLECTIVES") == "1" and "MASTER_PORT" in os.environ
    seen = devices = None
    if world > 1 or one_rank_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world and dist.get_rank() == rank
        seen, devices = ranks_seen(rank, local_rank, backend, dev)
        if sorted(seen) != list(range(world)):
            raise SystemExit(f"bench.py: the group holds ranks {seen}, expected 0..{world - 1}")
        if backend == "nccl" and world > 1 and len(set(devices)) != world:
            raise SystemExit(f"bench.py: RCCL ranks share devices {devices}: one GPU per rank is the contract")

    from fusionsense_amd import frame_cache, ops

    log('building workload')
    if args.express is not None:
        os.environ["FSGS_EXPRESS"] = str(args.express)  # (read by SplatTrainer at construction)
    if args.bin_fill is not None:
        import fusionsense_amd.fused as _f
        _f.BIN_FILL = bool(args.bin_fill)
    if args.bwd_order is not None:
        import fusionsense_amd.fused as _f
        _f.BWD_ORDER = bool(args.bwd_order)
    if args.sort_order is not None:
        ops.BIN_FILL_SORT_ORDER = bool(args.sort_order)
        ops.TWO_PASS_TILE_ORDER = bool(args.sort_order)
    if args.fwd_order is not None:
        import fusionsense_amd.fused as _f
        _f.FWD_TILE_ORDER = bool(args.fwd_order)
    if args.bwd_queue is not None:
        import fusionsense_amd.fused as _f
        _f.BWD_QUEUE_ITEMS = int(args.bwd_queue)
    if args.handoff_rel_len is not None:
        from fusionsense_amd.fused import FWD_WALK
        FWD_WALK.handoff_rel_len = int(args.handoff_rel_len)
    if args.handoff_gate_len is not None:
        from fusionsense_amd.fused import FWD_WALK
        FWD_WALK.handoff_gate_len = int(args.handoff_gate_len)
    if args.fwd_walk is not None or args.handoff is not None or args.tail_items is not None:  # A/B switches
        from fusionsense_amd.fused import FWD_WALK
        if args.tail_items is not None:
            FWD_WALK.max_items = int(args.tail_items)
        if args.fwd_walk is not None:
            FWD_WALK.forced, FWD_WALK.forced_walk = True, int(args.fwd_walk)
        if args.handoff is not None:
            FWD_WALK.handoff_records = int(args.handoff)
    wl = build_workload(args, dev)
    trainer, cams, targets, W, H = wl["trainer"], wl["cams"], wl["targets"], wl["W"], wl["H"]
    fused = not args.unfused_caller
    log('workload built')

    def view_of(step):
        return (step * world + rank) % len(cams)

    grouped = dist.is_initialized()

    def barrier():
        if grouped:
            dist.barrier()

    # ---- setup: size the workspace pool for every view (forward + backward, no optimizer step), so that
    # neither the warmup nor the timed steps meet a first-time device allocation ----
    strategy = trainer.strategy
    if not os.environ.get("FSGS_BENCH_NO_PRIME"):
        so = strategy.stats_only
        strategy.stats_only = True  # (no callbacks while priming)
        # (twice: a view's first visit is binned by the two-pass route, its revisits by the count-free one, whose bucket
        # room has to be in the pool as well; the forward's demand probe for tail workgroups also settles here)
        for v in list(range(len(cams))) * 2 + list(range(min(len(cams), 24))):
            trainer.train_step(cams[v], targets[v], optimizer_step=False)
        torch.cuda.synchronize()
        strategy.stats_only = so
        strategy.xys_grad_norm = strategy.vis_counts = strategy.max_2Dsize = None
        log('workspace primed')
    trainer.step = wl["start_step"] - args.warmup if wl["start_step"] else 0

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
    # ... and of its launches every fourth (runs of 40 steps and more): the two event records cost ~5 us of every step
    # they bracket (config #2: 2 017 it/s with all hundred launches timed, 2 042 with none)
    timer_every = 8 if args.steps >= 80 else (4 if args.steps >= 40 else 1)
    ops.TIMER.reset(enabled=not os.environ.get("FSGS_BENCH_NO_TIMER"), only=("raster_bwd",), prealloc=args.steps + 2,
                    every=timer_every)
    # keep the interpreter's cyclic collector out of the timed region (a generation-2 pass costs tens of ms).
    gcm = os.environ.get("FSGS_BENCH_GC", "fd")
    if "c" in gcm:
        gc.collect()
    if "f" in gcm:
        gc.freeze()
    if "d" in gcm:
        gc.disable()
    # host time of the schedule's callbacks (they synchronise on their own: mask counts decide allocation sizes)
    cb_ms = {"before_train": [], "maybe_refine": []}

    def timed_cb(name):
        fn = getattr(strategy, name)

        def wrapper(*a, **k):
            if strategy.stats_only or (name == "maybe_refine" and trainer.step % strategy.cfg.refine_every != 0) or \
                    (name == "before_train" and trainer.step != strategy.add_touch_at):
                return fn(*a, **k)
            # (no synchronisation added around the call: the window is timed as a trainer would run it; a callback's
            # own waits — its mask counts — are inside the figure, launches it leaves queued are not)
            t_ = time.perf_counter()
            r_ = fn(*a, **k)
            cb_ms[name].append(round((time.perf_counter() - t_) * 1e3, 2))
            return r_
        return wrapper

    if args.config == 3:
        strategy.before_train = timed_cb("before_train")
        strategy.maybe_refine = timed_cb("maybe_refine")
    if grouped:
        trainer.comm_events = []  # HIP-event pairs around every collective / wait of the timed steps (trainer._comm)
    torch.cuda.synchronize()
    n_alloc0 = torch.cuda.memory_stats(dev).get('num_device_alloc', 0)
    n_before = trainer.num_gaussians()
    tune0 = (bwd_dispatch_choice(dev, W, H) or {}).get("tuning_frames", 0)
    ftune0 = (fwd_walk_choice(dev, W, H, trainer.num_gaussians()) or {}).get("tuning_frames", 0)
    t0 = time.perf_counter()
    step_ends = []
    # one event per step on the step's stream: the GPU-side duration of every step, read after the timed region
    # (config #2's long default run: behind every fourth step only — see gpu_step_stats; FSGS_BENCH_STEP_EVENTS_EVERY moves it)
    ev_every = int(os.environ.get("FSGS_BENCH_STEP_EVENTS_EVERY", "4" if (args.config == 2 and args.steps >= 40
                                                                           and args.steps % 4 == 0 and not args.densify) else "1"))
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps // ev_every + 1)]
    step_views = []
    with_events = not os.environ.get("FSGS_BENCH_NO_STEP_EVENTS")
    if with_events:
        step_events[0].record()
    stall_calls = []
    if os.environ.get("FSGS_BENCH_STALL_TRACE"):
        # diagnostic: which C-ABI call (a launch: it should return in microseconds) blocks the host for over 1 ms
        from fusionsense_amd import _lib as _L
        lib_ = _L.load()

        def timed_entry(name, fn):
            def call(*a):
                t_ = time.perf_counter()
                r_ = fn(*a)
                dt = time.perf_counter() - t_
                if dt > 1e-3:
                    stall_calls.append((len(step_ends), name, round(dt * 1e3, 2)))
                return r_
            return call
        for name in _L.SIGNATURES:
            setattr(lib_, name, timed_entry(name, getattr(lib_, name)))
    host_issue = []  # per step: host time inside train_step MINUS the wait for the frame's verdict (one-call steps only)
    for s in range(args.steps):
        v = view_of(args.warmup + s)
        h0_ = time.perf_counter()
        x0_ = getattr(trainer, "express_steps", 0)
        trainer.train_step(cams[v], targets[v])
        if getattr(trainer, "express_steps", 0) > x0_:
            host_issue.append((time.perf_counter() - h0_) * 1e3 - trainer._express.last_wait_ms)
        if with_events and (s + 1) % ev_every == 0:
            step_events[(s + 1) // ev_every].record()
        step_views.append(v)
        step_ends.append(time.perf_counter())
    if stall_calls:
        log("C-ABI calls over 1 ms: " + json.dumps(stall_calls))
    trainer.flush()  # (data-parallel runs: the last step's deferred feature update belongs to the timed work)
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0  # (this rank's own clock, before it waits for the others)
    tune_in_region = (bwd_dispatch_choice(dev, W, H) or {}).get("tuning_frames", 0) - tune0
    ftune_in_region = (fwd_walk_choice(dev, W, H, trainer.num_gaussians()) or {}).get("tuning_frames", 0) - ftune0
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    gc.unfreeze()
    kernel_ms = ops.TIMER.summary()
    if os.environ.get("FSGS_BENCH_PARAM_DIGEST"):
        # (float atomics: two runs agree to rounding, not to the bit — a digest to compare with a tolerance)
        log('param digest ' + json.dumps({name: [float(trainer.params[name].data.double().mean()),
                                                 float(trainer.params[name].data.double().abs().mean())]
                                          for name in sorted(trainer.params)}))
    replicas_ok = None
    if grouped and os.environ.get("FSGS_BENCH_CHECK_REPLICAS", "1" if world > 1 else "") not in ("", "0"):
        # the ranks must hold bit-identical parameters after the timed steps (rank-ordered gradient sums)
        for name, prm in trainer.params.items():
            ref = prm.data.clone()
            dist.broadcast(ref, 0)
            assert torch.equal(ref, prm.data), f"rank {rank}: replica of {name} differs from rank 0"
        replicas_ok = True
        log('replicas identical')
    n_alloc = torch.cuda.memory_stats(dev).get('num_device_alloc', 0) - n_alloc0
    log(f'timed region done: {elapsed:.3f}s')
    ops.TIMER.reset(enabled=False)
    rank_stats = None
    if grouped:
        # per-rank step time and EXPOSED communication (time the step's stream spent inside / blocked by a collective:
        # what the deferred exchange did not hide), gathered so that a scaling curve can be read from the line
        comm_ms = sum(a.elapsed_time(b) for _, a, b in (trainer.comm_events or [])) / max(args.steps, 1)
        trainer.comm_events = None
        gs = gpu_step_stats(step_events, step_views, ev_every) if with_events else None
        mine = torch.tensor([own_elapsed / args.steps * 1e3, comm_ms, gs["p50"] if gs else float("nan")],
                            dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        per = [g.cpu().tolist() for g in got]
        sm, cm, gm = sorted(x[0] for x in per), sorted(x[1] for x in per), sorted(x[2] for x in per)
        rank_stats = {"step_ms_per_rank": {"min": round(sm[0], 4), "median": round(sm[len(sm) // 2], 4),
                                           "max": round(sm[-1], 4),
                                           "gpu_p50_min_median_max": [round(gm[0], 4), round(gm[len(gm) // 2], 4),
                                                                      round(gm[-1], 4)]},
                      "comm_ms_exposed": {"min": round(cm[0], 4), "median": round(cm[len(cm) // 2], 4),
                                          "max": round(cm[-1], 4),
                                          "note": "per step and rank: HIP-event time of the step's stream inside or "
                                                  "blocked by collectives (reduce / reduce-scatter / all-gather and the "
                                                  "wait for the deferred feature exchange)"}}
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()

    # ---- side measurements (outside the timed region; no callbacks, so the scene stays as the timed region left it) ----
    strategy.stats_only = True
    side_steps = min(args.steps, 10)
    ops.TIMER.reset(enabled=True)
    for s in range(side_steps):
        v = view_of(s)
        _, trainer._bench_last_out = trainer.train_step(cams[v], targets[v])
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
    # L: entries of its tile's sorted live list a pixel walks up to and including the last one it
    # composites (last_ids - the tile's list start + 1; 0 for pixels nothing reached), mean over the frame's pixels
    with torch.no_grad():
        li = info["last_ids"].reshape(H, W).long()
        offs = info["isect_offsets"].reshape(-1)[: ((H + 15) // 16) * ((W + 15) // 16)].long()
        ty = torch.arange(H, device=dev) // 16
        tx = torch.arange(W, device=dev) // 16
        start = offs[(ty[:, None] * ((W + 15) // 16) + tx[None, :])]
        acc_img = trainer_last_alpha(trainer)
        walked = torch.where(acc_img.reshape(H, W) > 0, li - start + 1, torch.zeros_like(li)).clamp_min(0)
        mean_walked = float(walked.float().mean())
        max_walked = int(walked.max())
    side_n = min(args.steps, 50)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for s in range(side_n):
        v = view_of(s)
        trainer.train_step(cams[v], targets[v], optimizer_step=False)
    torch.cuda.synchronize()
    t_noopt = (time.perf_counter() - t1) / side_n
    # forward-only latency -> rendered Mpix/s (num_rays_per_sec of dn_pipeline.py:246-248)
    with torch.no_grad():
        for s in range(min(side_n, len(cams))):  # (every view once: the workspace of the no-grad path grows on first use)
            trainer.forward(cams[view_of(s)])
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for s in range(side_n):
            trainer.forward(cams[view_of(s)])
        torch.cuda.synchronize()
        t_fwd = (time.perf_counter() - t2) / side_n

    # the same steps through the drop-in surface (rasterization() + rasterize_gaussians(), the reference's own call
    # structure): what a FusionSense maintainer gets without touching dn_model.py
    dropin = None
    if fused and not args.no_dropin and args.config in (2, 3) and world == 1:
        from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
        from fusionsense_amd.trainer import SplatTrainer
        d_st = DensifyStrategy(SplatfactoConfig(), num_train_data=len(cams), stats_only=True)
        d_st.add_mask, d_st.touch_normals = strategy.add_mask, strategy.touch_normals
        d_tr = SplatTrainer({k: p.data for k, p in trainer.params.items()}, dev, sh_degree=3, fused=False,
                            loss_cfg=trainer.loss_cfg, strategy=d_st)
        d_tr.step = trainer.step
        d_tr.sh_degree_interval = trainer.sh_degree_interval
        from fusionsense_amd import rendering
        for s in range(5):
            d_tr.train_step(cams[view_of(s)], targets[view_of(s)])
        torch.cuda.synchronize()
        # the rate first, UNINSTRUMENTED (per-launch HIP events and clock reads cost this host-bound route ~25 %: round 3
        # quoted the instrumented figure) ...
        nd = 30
        t3 = time.perf_counter()
        for s in range(nd):
            d_tr.train_step(cams[view_of(s)], targets[view_of(s)])
        torch.cuda.synchronize()
        dropin = nd / (time.perf_counter() - t3)
        # ... then the same steps again with the library's host time and kernel time taken
        for k_ in rendering.HOST_TIME:
            rendering.HOST_TIME[k_] = 0
        rendering.MEASURE_HOST_TIME = True  # (clock reads inside the two calls only for this side measurement)
        ops.TIMER.reset(enabled=True)
        t3i = time.perf_counter()
        for s in range(nd):
            d_tr.train_step(cams[view_of(s)], targets[view_of(s)])
        torch.cuda.synchronize()
        dropin_instrumented = nd / (time.perf_counter() - t3i)
        d_k = ops.TIMER.summary()
        ops.TIMER.reset(enabled=False)
        ht = rendering.HOST_TIME
        rendering.MEASURE_HOST_TIME = False
        # library side of the drop-in route per step: host time inside rasterization() + rasterize_gaussians()
        # (forward calls; includes the one wait for the live-pair count) against the GPU time of the library's kernels
        dropin_detail = {
            "host_ms_in_rasterization_per_call": round(1e3 * ht["rasterization_s"] / max(ht["rasterization_calls"], 1), 4),
            "host_ms_in_rasterize_gaussians_per_call": round(1e3 * ht["rasterize_gaussians_s"] / max(ht["rasterize_gaussians_calls"], 1), 4),
            "library_kernels_gpu_ms_per_step": round(sum(v["avg_ms"] * v["calls"] for v in d_k.values()) / nd, 4),
            "ms_per_step": round(1e3 / dropin, 3),
            "iters_per_s_while_instrumented": round(dropin_instrumented, 2),
            "note": "the rest of the step is the reference's own torch glue, autograd and optimizers, untouched; the two "
                    "host times and the kernel time are from the instrumented pass, ms_per_step from the plain one",
        }
        del d_tr
        # integration.patch(): get_outputs on the fused node, but still on torch's autograd tape and under
        # torch.optim.Adam objects stepped one by one, as nerfstudio's trainer does
        p_tr = SplatTrainer({k: p.data for k, p in trainer.params.items()}, dev, sh_degree=3, fused=True, direct=False,
                            loss_cfg=trainer.loss_cfg, strategy=d_st, torch_optimizers=True)
        p_tr.step = trainer.step
        p_tr.sh_degree_interval = trainer.sh_degree_interval
        for s in range(5):
            p_tr.train_step(cams[view_of(s)], targets[view_of(s)])
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for s in range(nd):
            p_tr.train_step(cams[view_of(s)], targets[view_of(s)])
        torch.cuda.synchronize()
        dropin_detail["patched_get_outputs_iters_per_s"] = round(nd / (time.perf_counter() - t4), 2)
        del p_tr
        # integration.patch_all(): get_outputs AND get_loss_dict as HIP autograd nodes, the statistics applied by the
        # backward, DensifyStrategy behind the callbacks, and the torch.optim.Adam objects stepped one by one as
        # nerfstudio's Optimizers does with integration.AdamFuser gathering them into one libfsgs launch — the
        # reference's loop shape (torch's tape, optimizer objects, callbacks), every kernel on the HIP path
        f_tr = SplatTrainer({k: p.data for k, p in trainer.params.items()}, dev, sh_degree=3, fused=True, direct=False,
                            loss_cfg=trainer.loss_cfg, strategy=d_st, torch_optimizers=True, fuse_torch_optimizers=True)
        f_tr.step = trainer.step
        f_tr.sh_degree_interval = trainer.sh_degree_interval
        for s in range(5):
            f_tr.train_step(cams[view_of(s)], targets[view_of(s)])
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for s in range(nd):
            f_tr.train_step(cams[view_of(s)], targets[view_of(s)])
        torch.cuda.synchronize()
        patched_full = round(nd / (time.perf_counter() - t5), 2)
        dropin_detail["patched_full_iters_per_s"] = patched_full
        del f_tr

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        iters_per_s = world * args.steps / elapsed
        # algorithmic bytes.  The normal plane reuses the RGB+ED pass's sorted
        # lists when both bbox rules bin the frame identically (+80 B/isect), else re-bins (+244).
        key_bits = 32 + ops.tile_bits((W // 16 + (W % 16 > 0)) * (H // 16 + (H % 16 > 0)))
        sort_b = 2 * 12 * ((key_bits + 7) // 8)
        b_isect = 12 + sort_b + 8 + 44 + 44
        reused = fused or (frame_cache.hits > 0 and frame_cache.misses == 0)
        b_isect_normal = (40 + 40) if reused else (12 + sort_b + 8 + 40 + 40)
        b_iter_rect = N * 352 + n_vis * 444 + M * (b_isect + b_isect_normal) + P * 92
        # dominant kernel = the libfsgs launch with the largest share of the step; algorithmic bytes
        # per launch from the per-unit figures of SURVEY.md §8d
        # (frames binned with occlusion cuts hand the compositing kernels a depth prefix of the lists: their unit is then
        # the pairs that were binned at all, not gsplat's rectangle count, which would price them at > 1 of the peak)
        cuts_on = int(getattr(trainer, "cut_frames", 0)) > 0
        M_r = n_live if cuts_on else M
        b_iter = N * 352 + n_vis * 444 + M_r * (b_isect + b_isect_normal) + P * 92
        r_note = " [unit: binned pairs — occlusion cuts]" if cuts_on else ""
        alg = {
            "raster_bwd_quad_d4e3": ("raster_bwd_live_kernel<4,true,3> (RGB+ED and normal plane, one walk)" + r_note,
                                     M_r * (44 + 40) + P * (28 + 20) + n_vis * (48 + 28)),
            "raster_fwd_quad_d4e3": (("raster_fwd_wave_kernel<4,3> (one wave per quadrant, long walks "
                                      "finished chunk-parallel by tail workgroups of the same launch" if (fwd_walk_choice(dev, W, H, trainer.num_gaussians()) or {}).get("handoff_records")
                                      else "raster_fwd_wave_kernel<4,3> (one wave per quadrant" if (fwd_walk_choice(dev, W, H, trainer.num_gaussians()) or {}).get("walk") == 1
                                      else "raster_fwd_quad_kernel<4,3> (four waves per quadrant") +
                                     ": filter + gather + composite, RGB+ED and normal plane)" + r_note,
                                     M_r * (44 + 40) + P * (24 + 20)),
            "raster_bwd_quad_d4": ("raster_bwd_live_kernel<4,true,0>", M * 44 + P * 28 + n_vis * 48),
            "raster_bwd_quad_d3": ("raster_bwd_live_kernel<3,false,0>", M * 40 + P * 20 + n_vis * 28),
            "raster_fwd_quad_d4": ("raster_fwd_quad_kernel<4,0>", M * 44 + P * 24),
            "raster_fwd_quad_d3": ("raster_fwd_quad_kernel<3,0>", M * 40 + P * 20),
            "raster_bwd_d4": ("raster_bwd_kernel<4,true>", M * 44 + P * 28 + n_vis * 48),
            "sort_pairs": ("radix sort (hist + scan + scatter per 8-bit pass)", M * sort_b),
            # bucket fill + in-tile sorts (direct binning): per LIVE pair the fill pass writes one 8-byte word into its
            # tile's bucket, the sort reads it once and writes the 4-byte payload (the word never leaves LDS / registers
            # in between); both enumeration passes read 28 B per Gaussian (centre, radius, conic, opacity, depth)
            "tile_sort": ("isect_live_bin<fill> + tile_sort_kernel2 (bucket fill + per-tile LDS sort), priced on what "
                          "it moves: 8 B written + 8 B read + 4 B written per live pair, 28 B per Gaussian",
                          n_live * 20 + N * 28),
            "adam_step": ("adam_kernel (six parameter groups, one launch)", N * 59 * 28),
        }
        riding = bool(getattr(trainer, "_ride_mode", lambda: False)())
        aib_on = int(getattr(trainer, "adam_in_backward_steps", 0)) > 0
        idle_frac = never_updated_frac(trainer) or 0.0
        alg.update({
            "isect_count_live": ("projecting count pass + table scan with the SH colours / record packing riding"
                                 + (" + the features' Adam step riding" if riding else ""),
                                 N * (68 + 20 + 16) + n_vis * (216 + 64) + (N * 48 * 28 if riding else 0)),
            # (with the step's Adam applied in this launch — trainer.adam_in_backward_steps — the 236 B of gradients
            # are not written; parameters and both moments are read and written instead: 59 floats x 6 streams)
            "gaussian_bwd": ("gauss_sh_bwd_kernel (SH VJP + projection / normal / activation VJPs + after_train statistics"
                             + (" + the Adam step of all six groups" if aib_on else "") + ")",
                             # (... of which a float with zero gradient on zero moments only has its moments read: 8 B)
                             N * (120 + 52 + 36 + 76 + ((59 * 8 - 236) if aib_on else 0)) + n_vis * 228
                             + (int((1.0 - idle_frac) * N) * 59 * 16 if aib_on else 0)),
            "ssim_l1_fwd": ("ssim_l1_fwd_kernel", P * 3 * 24),
            "ssim_l1_bwd": ("ssim_l1_bwd_kernel (+ loss combine)", P * 3 * 36),
        })

        def roofline_of(key):
            """Roofline block of one libfsgs span: algorithmic bytes (SURVEY.md §8d per-unit figures x this frame's
            units) over the span's average duration from HIP events of THIS run; HBM traffic and vector-ALU counters
            from this round's committed rocprofv3 --pmc passes of the same command (profiles/pmc_traffic.json).  The
            limiter is DERIVED from those: vector ALUs >= 65 % busy -> VALU issue; algorithmic or counted traffic
            >= 60 % of the HBM peak -> HBM; neither -> latency / memory-side atomics (DESIGN.md §9 cites the ablation)."""
            name, nbytes = alg[key]
            ms = kernel_ms[key]["avg_ms"]
            ach = nbytes / (ms * 1e-3) / 1e9
            # (the per-Gaussian backward that also applies the Adam step is an instantiation of its own: own counters)
            rec = pmc_record("gaussian_bwd_adam" if (key == "gaussian_bwd" and aib_on) else key, args.config)
            r = {"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(ach / HBM_PEAK_GBS, 5),
                 # rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE (bytes per launch, gfx950-corrected) of this kernel
                 # from this round's committed run in profiles/ (DESIGN.md §5), null if absent
                 "traffic": rec.get("hbm_bytes_per_launch"), "traffic_source": rec.get("source"),
                 "algorithmic_bytes": nbytes, "avg_launch_ms": round(ms, 4),
                 "launches_timed": kernel_ms[key]["calls"]}
            traffic_frac = (rec["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                            if rec.get("hbm_bytes_per_launch") else None)
            if traffic_frac is not None:
                r["traffic_frac_of_peak"] = round(traffic_frac, 4)
                r["write_bytes_per_launch"] = rec.get("write_bytes_per_launch")
            q = rec.get("sq_active_inst_valu_quadcycles_per_launch")
            n_valu = rec.get("sq_insts_valu_per_launch")
            issue = None
            if q and n_valu:
                # The vector-issue roofline beside the HBM one, on MEASURED rates (profiles/r6_valu_rate.txt, tools/valu_rate.hip):
                # SQ_INSTS_VALU counts wave instructions; SQ_ACTIVE_INST_VALU counts 4-cycle PASSES as if nothing paired
                # (1 per ordinary instruction, 2 per transcendental / v_permlane*_swap).  A SIMD retires a DPP / compare /
                # select / min-max class instruction in 4.2 cycles, a transcendental in 8.2, and a simple fp32 / integer
                # instruction (v_fma / mul / add / sub / mov / and / or / add_u32) in 2.2 when a second wave's instruction
                # shares the pass — never when that is a DPP or transcendental one — else 4.4.  The static class shares of the
                # kernel's hot loop (tools/isa_mix.py -> profiles/r6_isa_mix.json) split the counted instructions.
                mix = isa_mix_record(key)
                n_trans = max(q - n_valu, 0)
                n_rest = max(n_valu - n_trans, 0)
                sh_simple, sh_dpp = mix.get("simple_share", 0.0), mix.get("dpp_share", 0.0)
                n_simple, n_dpp = n_rest * sh_simple, n_rest * sh_dpp
                n_full = n_rest - n_simple - n_dpp
                # best case: every non-DPP 4-cycle instruction shares its pass with a simple one, the remaining simple ones pair up
                passes_best = max((n_simple + n_full) / 2.0, n_full) + n_dpp
                cyc_best = 4.3 * passes_best + 8.2 * n_trans
                cyc_unpaired = 4.3 * n_rest + 8.2 * n_trans
                simd_hz = 1024 * 2.4e9
                floor_ms = cyc_best / simd_hz * 1e3
                unpaired_ms = cyc_unpaired / simd_hz * 1e3
                issue = min(unpaired_ms / ms, 1.0)  # share of the launch the SIMDs issue vector work if NOTHING paired
                r["valu_insts_per_launch"] = n_valu
                r["valu_roofline"] = {
                    "bound": "valu", "insts_per_launch": n_valu, "transcendental_or_permlane_swap": int(n_trans),
                    "hot_loop_mix": mix or None, "cycles": {"simple_paired": 2.2, "simple_alone": 4.4, "full": 4.2,
                                                            "transcendental": 8.2},
                    "simds": 1024, "clock_ghz": 2.4, "floor_ms": round(floor_ms, 4), "frac": round(min(floor_ms / ms, 1.0), 4),
                    "issue_ms_if_nothing_paired": round(unpaired_ms, 4),
                    "calibration": "profiles/r6_valu_rate.txt",
                    "note": ("floor = every simple instruction shares its 4-cycle pass (the best the issue logic was seen to do); "
                             "issue_ms_if_nothing_paired = the same instructions one per pass: the launch lies between the two")}
                # (round 5 reported 4 x SQ_ACTIVE_INST_VALU / SIMD cycles as 'valu_busy': it exceeds 1 when simple
                # instructions pair, because the counter charges them a full pass each — kept only as the raw count)
                r["sq_active_inst_valu_passes_per_launch"] = q
            wr = rec.get("write_bytes_per_launch")
            if wr and key.startswith("raster_bwd"):
                # memory-side atomics: every (instruction, 64-byte line) of a global_atomic_add_f32 is one operation that
                # WRITE_SIZE charges 64 B, and the chip retires 20.2 G of them per second whatever the footprint
                ops = wr / 64.0
                a_ms = ops / 20.2e9 * 1e3
                r["atomic_roofline"] = {"bound": "memory-side atomics", "line_ops_per_launch": int(ops), "peak_gops": 20.2,
                                        "floor_ms": round(a_ms, 4), "frac": round(min(a_ms / ms, 1.0), 4),
                                        "calibration": "profiles/r6_valu_rate.txt"}
            if issue is None and traffic_frac is None:
                r["limiter"] = None  # no counters committed for this span and configuration
            elif issue is not None and issue >= 0.65:
                r["limiter"] = "vector-ALU issue (>= 0.65 of the launch if no two instructions shared a pass)"
            elif max(ach / HBM_PEAK_GBS, traffic_frac or 0.0) >= 0.6:
                r["limiter"] = "HBM bandwidth (>= 0.6 of the peak)"
            else:
                r["limiter"] = "latency / memory-side atomics (vector issue < 0.65, HBM < 0.6 of the peak)"
            return r

        cand = sorted(((v["avg_ms"], k) for k, v in kernel_ms.items() if k in alg), reverse=True)  # (one launch per step each)
        roofline = roofline_of(cand[0][1]) if cand else None
        roofline_top3 = [roofline_of(k) for _, k in cand[:3]]
        line = {
            "metric": "train_iters_per_s",
            "value": round(iters_per_s, 3),
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.config != 5 else "f32 arithmetic, f16 attribute storage",
            "data": "synthetic",
            "config": {"workload": wl["name"] + "; 1 view/rank/step; caller = " +
                                   ("fused get_outputs node" if fused else
                                    "reference-style op-by-op glue over the drop-in surface"),
                       "baseline_config": args.config, "n_gaussians": N, "n_gaussians_start": n_before,
                       "width": W, "height": H, "views": len(cams),
                       "n_isects": M, "n_isects_live": n_live, "n_visible": n_vis,
                       "isects_per_gaussian": round(M / max(N, 1), 3),
                       # L of SURVEY.md §8d: list entries a pixel walks up to its last composited one (frame mean / max)
                       "mean_walked_list": round(mean_walked, 2), "max_walked_list": max_walked,
                       # live (binned) pairs over gsplat's rectangle pairs: reach test + (dense scenes) occlusion cuts
                       "binned_over_rect_pairs": round(n_live / max(M, 1), 4),
                       "scene": wl.get("scene"),
                       "normal_pass_list_reuse": bool(reused), "parallelism": f"dp{world}",
                       "backend": (backend if grouped else None),
                       "world_size": world, "ranks_seen": seen, "rank_devices": devices,
                       "sharded_geometry": getattr(trainer, "_geo", None) is not None,
                       "replicas_identical_after_timed_steps": replicas_ok,
                       "launched_by": ("bench.py (child torch.distributed.run)" if os.environ.get("FSGS_BENCH_SPAWNED")
                                       else ("external launcher" if "WORLD_SIZE" in os.environ else "single process")),
                       "comm_bytes_per_step_per_rank": getattr(trainer, "comm_bytes_last_step", 0) if grouped else 0},
            # row (e) of SURVEY.md 8: this line is ONE point of the 1 / 2 / 4 / 8-GPU curve the driver measures; no N > 1 RCCL
            # run exists in this repository's own records (the development pool has one GPU per box) — the efficiencies
            # in DESIGN.md 6 are sized from link figures, not measured
            "multi_gpu_curve": ("this line is one point of it" if world > 1 else
                                "unmeasured by the builder (1-GPU pool): see the driver's SCALE record"),
            "rendered_mpix_per_s": round(world * P / t_fwd / 1e6, 2),
            "fwd_ms": round(t_fwd * 1e3, 3),
            "iters_per_s_excl_optimizer": round(world / t_noopt, 3),
            "dropin_iters_per_s": None if dropin is None else round(dropin, 2),
            "dropin_detail": None if dropin is None else dropin_detail,
            "patched_full_iters_per_s": None if dropin is None else dropin_detail.get("patched_full_iters_per_s"),
            "render_tolerance": RENDER_TOLERANCE,
            # what the parity tests compare against: the rasterizer arithmetic lives in gsplat 1.0.0 /
            # nerfstudio 1.1.3, both absent from /root/reference — the oracle restates them and nothing reference-held pins it
            "oracle": "parity unpinned (gsplat 1.0.0 / nerfstudio 1.1.3 absent from the reference tree; the FusionSense-owned "
                      "glue, losses, densification and data formats ARE pinned by goldens produced by executing the reference)",
            "iter_algorithmic_bytes": b_iter,
            # (frames binned with occlusion cuts: priced on the pairs that were binned — the work the step does — so
            # that the fraction stays a fraction; the §8d figure on gsplat's rectangle pairs is kept beside it)
            "iter_hbm_frac": round(b_iter / (elapsed / args.steps) / (HBM_PEAK_GBS * 1e9), 5),
            "iter_algorithmic_bytes_rect_pairs": b_iter_rect,
            "device_mallocs_in_timed_region": n_alloc,
            "runtime_env": dict(RUNTIME_ENV),
            "live_list_overflows": int(getattr(trainer, "live_overflows", 0)),  # frames redone (no-wait binning)
            # frames by binning route over the whole run: "fill" = the count-free route (one enumeration of the pairs into
            # buckets sized from the view's previous frame), "two_pass" = count + fill (first visits, after densification,
            # dense scenes, redone frames)
            "binning_frames": dict(getattr(trainer, "bin_frames", {})),
            # steps of the timed region enqueued by ONE library call (fsgs_step_run; fusionsense_amd/express.py), and what
            # issuing such a step costs the host: time inside train_step minus the wait for the frame's verdict
            "express_steps_total": getattr(trainer, "express_steps", 0),
            "express_steps_timed": len(host_issue),
            "host_issue_ms_per_step": ({"p50": round(sorted(host_issue)[len(host_issue) // 2], 4),
                                        "p99": round(sorted(host_issue)[min(len(host_issue) - 1, int(0.99 * len(host_issue)))], 4),
                                        "max": round(max(host_issue), 4)} if host_issue else None),
            # dispatch order of the compositing backward: a rule of the frame's size since round 5 (fused._BwdDispatch)
            "bwd_dispatch": bwd_dispatch_choice(dev, W, H),
            "bwd_dispatch_tuning_frames_in_timed_region": tune_in_region,
            # the forward compositing's walk (fixed since round 5 — no tuner, reproducible runs: fused.FWD_WALK; the
            # ``walk`` / ``handoff_records`` arguments of fsgs_raster_fwd_quad)
            "fwd_walk": fwd_walk_choice(dev, W, H, trainer.num_gaussians()),
            "fwd_walk_tuning_frames_in_timed_region": ftune_in_region,
            # steps whose Adam update was applied inside the per-Gaussian backward launch (no gradient slab, no Adam
            # launch: DESIGN.md §9.9; FSGS_ADAM_IN_BACKWARD=auto|1|0)
            "adam_in_backward_steps": int(getattr(trainer, "adam_in_backward_steps", 0)),
            # Gaussians whose opacity logit has never received a gradient (second moment still exactly 0) after the run:
            # the Adam step inside the backward neither reads their parameters nor writes anything for them (an exact
            # no-op).  In this synthetic scene that is every Gaussian no view has reached yet — a real capture has fewer.
            "never_updated_gaussians_frac": never_updated_frac(trainer),
            # occlusion cuts (dense scenes): frames binned with cuts / of those, redone uncut because a
            # cut tile did not saturate (both over warmup + timed steps; the redone frames are inside the timed region)
            "occlusion_cut": {"mode": getattr(trainer, "occlusion_cut_mode", "0"),
                              "cut_frames": int(getattr(trainer, "cut_frames", 0)),
                              "cut_redone": int(getattr(trainer, "cut_redone", 0)),
                              "second_looks": int(ops.zcut_rechecks),
                              "margins_span_rel_tail": list(getattr(trainer, "zcut_margins", ()))},
            "max_step_ms": round(1e3 * max(b - a for a, b in zip([t0] + step_ends[:-1], step_ends)), 3),
            "slowest_step": max(range(len(step_ends)), key=lambda i: step_ends[i] - ([t0] + step_ends)[i]),
            # host-side duration (ms) of the steps around the slowest one, and the view it rendered
            "slowest_step_view": step_views[max(range(len(step_ends)), key=lambda i: step_ends[i] - ([t0] + step_ends)[i])],
            "host_ms_around_slowest": (lambda d, i: [round(1e3 * x, 3) for x in d[max(i - 3, 0):i + 4]])(
                [b - a for a, b in zip([t0] + step_ends[:-1], step_ends)],
                max(range(len(step_ends)), key=lambda i: step_ends[i] - ([t0] + step_ends)[i])),
            "gpu_step_ms": gpu_step_stats(step_events, step_views, ev_every) if with_events else None,
            "kernels_ms": kernel_ms,
            "roofline": roofline,
            "roofline_top3": roofline_top3,
        }
        if rank_stats is not None:
            line.update(rank_stats)
        if args.config == 3:
            line["config"]["refinement"] = dict(strategy.last_report)
            line["config"]["touch_anchors"] = int(strategy.add_mask.sum()) if strategy.add_mask is not None else 0
            cb_total = sum(cb_ms["before_train"]) + sum(cb_ms["maybe_refine"])
            line["callbacks_ms"] = {"add_touch_patch": cb_ms["before_train"], "refinement_hull_touch_pruning": cb_ms["maybe_refine"]}
            line["iters_per_s_excl_callbacks"] = round(world * args.steps / max(elapsed - cb_total * 1e-3, 1e-9), 3)
        log('gpu part done')
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
