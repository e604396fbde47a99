"""Shared builders for the parity tests."""
import math

import numpy as np
import torch

from fusionsense_amd import scenes
from oracle import gsplat_ref as R


def activated(params):
    """Stored parametrisation -> the activated tensors rasterization() receives
    (dn_splatter/dn_model.py:572-575, 543)."""
    q = params["quats"]
    return dict(
        means=params["means"],
        quats=q / q.norm(dim=-1, keepdim=True),
        scales=torch.exp(params["scales"]),
        opacities=torch.sigmoid(params["opacities"]).squeeze(-1),
        colors=torch.cat((params["features_dc"][:, None, :], params["features_rest"]), dim=1),
    )


def camera_mats(cam):
    return R.get_viewmat(cam.c2w[None]), cam.K()[None]


def adversarial_gaussians(cam, seed=1):
    """64 hand-picked Gaussians around config #1's camera: behind the camera, beyond far,
    off-screen, huge (clamp active), tiny, degenerate scales, on tile boundaries."""
    g = torch.Generator().manual_seed(seed)
    n = 64
    means = torch.rand(n, 3, generator=g) - 0.5
    scales = torch.exp(math.log(0.01) + (math.log(0.2) - math.log(0.01)) * torch.rand(n, 3, generator=g))
    quats = scenes.random_quat_tensor(n, g)
    means[0] = torch.tensor([0.0, 0.0, 3.0])      # behind the camera (camera at z=2.5 looking -z)
    means[1] = torch.tensor([0.0, 0.0, 2.495])    # closer than near plane
    means[2] = torch.tensor([50.0, 0.0, 0.0])     # far off-screen right
    means[3] = torch.tensor([0.0, -60.0, 0.0])    # far off-screen
    means[4] = torch.tensor([1.2, 1.2, 2.0])      # x/z beyond 1.3*tan(fov): clamp active
    scales[4] = torch.tensor([0.5, 0.5, 0.5])
    means[5] = torch.tensor([-1.0, 1.1, 2.2])
    scales[5] = torch.tensor([0.3, 0.01, 0.3])
    scales[6] = torch.tensor([1e-6, 1e-6, 1e-6])  # tiny -> eps2d dominates
    scales[7] = torch.tensor([2.0, 2.0, 2.0])     # covers the whole image
    means[7] = torch.zeros(3)
    scales[8] = torch.tensor([0.0, 0.0, 0.0])     # exactly degenerate
    quats[9] = torch.tensor([2.0, 0.0, 0.0, 0.0])  # non-unit quaternion
    quats[10] = quats[10] * 1e-3
    # exact pixel-grid placements: z=0 plane, fx=128 at distance 2.5 -> 51.2 px per unit
    for k in range(11, 20):
        means[k] = torch.tensor([(k - 15) * 16.0 / 51.2, 0.0, 0.0])
    opac = torch.rand(n, generator=g)
    return means, quats, scales, opac


def oracle_frame(act, viewmat, K, W, H, tile=16):
    """Oracle projection + binning for one camera; returns dict of tensors/arrays."""
    radii, m2, dp, cn, comp = R.project(act["means"], act["quats"], act["scales"], viewmat, K, W, H)
    tw, th = math.ceil(W / tile), math.ceil(H / tile)
    tpg, ids, flat = R.isect_tiles(m2, radii, dp, tile, tw, th)
    ids_s, flat_s = R.sort_isects(ids, flat)
    offs = R.isect_offset_encode(ids_s, viewmat.shape[0], tw, th)
    return dict(radii=radii, means2d=m2, depths=dp, conics=cn, comp=comp, tpg=tpg, ids=ids, flat=flat,
                ids_s=ids_s, flat_s=flat_s, offsets=offs, tw=tw, th=th)


def poison_capacity(caps, pairs):
    """The next estimate a KNOWN frame shape asks ``caps`` (a capacity.LiveCapacity) for is ``pairs`` (no margin), so that
    frame overflows its lists; the estimate after it is the real one again.  (A test hook: it patches the instance.)"""
    real = caps.get

    def once(key):
        if real(key) == 0:
            return 0
        del caps.get  # (the instance attribute: the class's method is back)
        return int(pairs)
    caps.get = once


def device_kernels_of(fn):
    """Names of the device kernels (and copies / memsets) ``fn()`` launches, from torch.profiler's device activity."""
    from torch.profiler import ProfilerActivity, profile
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = []
    for e in prof.events():
        if str(getattr(e, "device_type", "")).endswith("CUDA") and e.name:
            names.append(e.name)
    return names


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def crop_camera(cam0, crop_w, crop_h=None, shift=(0, 0)):
    """The window of ``cam0``'s image of size crop_w x crop_h centred at the image centre + ``shift`` (pixels), as a
    camera of its own: same pose and focal lengths, shifted principal point."""
    from fusionsense_amd.scenes import Camera
    crop_h = crop_h or crop_w
    x0 = (cam0.width - crop_w) // 2 + shift[0]
    y0 = (cam0.height - crop_h) // 2 + shift[1]
    return Camera(cam0.c2w, cam0.fx, cam0.fy, cam0.cx - x0, cam0.cy - y0, crop_w, crop_h)


class OracleWindow:
    """The CPU oracle's frame of ALL of ``params_cpu``'s Gaussians through the (small) camera ``cam`` — forward once
    (fp32, graph kept), the fp32 / fp64 projections that decide which integer mismatches are rounding edges — against
    which any number of HIP routes are then checked (``check``): forward images, the integer outputs (radii,
    tiles_per_gauss) and every parameter gradient of a loss.

    No blanket allowance: every radii / tiles_per_gauss mismatch must be a Gaussian on which the fp32 and the fp64 oracle
    themselves disagree or whose fp64 pre-ceil radius / rectangle bound / cull bound sits within 1e-3 of an integer, and
    every pixel beyond the image tolerance must, walked in fp64, hold an entry on a threshold decision (alpha within 2e-6
    of 1/255 or of the 0.999 cap, running transmittance within 2e-7 of 1e-4)."""

    def __init__(self, params_cpu, cam, seed=11):
        from oracle.fusion_ref import render_fusionsense as render_ref
        self.cam = cam
        W, H = cam.width, cam.height
        g = torch.Generator().manual_seed(seed)
        self.w = {"rgb": torch.rand(H, W, 3, generator=g), "depth": torch.rand(H, W, 1, generator=g),
                  "normal": torch.rand(H, W, 3, generator=g), "accumulation": torch.rand(H, W, 1, generator=g)}
        self.pr = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in params_cpu.items()}
        self.orf = render_ref(self.pr, cam, sh_degree=3)
        act = activated({k: v.detach() for k, v in self.pr.items() if k not in ("features_dc", "features_rest")} |
                        {"features_dc": torch.zeros(1, 3), "features_rest": torch.zeros(1, 0, 3)})
        viewmat, K = camera_mats(cam)
        p32 = R.project(act["means"], act["quats"], act["scales"], viewmat, K, W, H)
        p64 = R.project(act["means"].double(), act["quats"].double(), act["scales"].double(), viewmat.double(),
                        K.double(), W, H)
        assert torch.equal(p32[0][0], self.orf["radii"])  # (the oracle run above is the fp32 oracle)
        # fp64 pre-ceil radius, recomputed from the fp64 conic (conic = inverse of the blurred 2-D covariance)
        cn = p64[3][0]
        detc = (cn[:, 0] * cn[:, 2] - cn[:, 1] ** 2).clamp(min=1e-300)
        a_, c_ = cn[:, 2] / detc, cn[:, 0] / detc
        det = 1.0 / detc
        bb = 0.5 * (a_ + c_)
        pre = 3.0 * torch.sqrt(bb + torch.sqrt(torch.clamp(bb * bb - det, min=0.01)))
        edge_r = ((pre - torch.round(pre)).abs() < 1e-3) | (p32[0][0] != p64[0][0])
        # culling / bbox edges: centre +- radius against the image borders within 1e-3 px
        mx, my, rr = p64[1][0][:, 0], p64[1][0][:, 1], p64[0][0].double()
        edge_cull = (((mx + rr).abs() < 1e-3) | ((mx - rr - W).abs() < 1e-3) | ((my + rr).abs() < 1e-3) |
                     ((my - rr - H).abs() < 1e-3))
        self.edge_radius = edge_r | edge_cull
        # tile rectangles: bounds (mean / 16 -+ radius / 16) within 1e-3 of an integer in fp64, or an unstable radius
        lo = torch.stack([mx, my], -1) / 16.0 - rr[:, None] / 16.0
        hi = torch.stack([mx, my], -1) / 16.0 + rr[:, None] / 16.0
        edge_t = (((lo - torch.round(lo)).abs() < 1e-3) | ((hi - torch.round(hi)).abs() < 1e-3)).any(-1)
        self.tw, self.th = (W + 15) // 16, (H + 15) // 16
        t32 = R.isect_tiles(p32[1], p32[0], p32[2], 16, self.tw, self.th)[0]
        t64 = R.isect_tiles(p64[1], p64[0], p64[2], 16, self.tw, self.th)[0]
        self.edge_tiles = (edge_t | torch.from_numpy(np.asarray(t32).reshape(-1) != np.asarray(t64).reshape(-1)) |
                           self.edge_radius)
        info = self.orf["info"]
        self.offs = np.concatenate([np.asarray(info["isect_offsets"]).reshape(-1).astype(np.int64),
                                    [len(info["flatten_ids"])]])
        self.flat = np.asarray(info["flatten_ids"]).astype(np.int64)
        self.m2_64, self.cn_64 = p64[1][0], p64[3][0]
        self.op_64 = act["opacities"].double().reshape(-1)
        self._oracle_grads = {}  # loss key -> (oracle loss value, oracle parameter gradients): see check()

    def loss_of(self, out, to):
        """The default loss: seeded random weights on every output image."""
        return sum((out[k] * to(self.w[k])).mean() for k in self.w)

    def _on_a_threshold(self, y, x):
        t = (y // 16) * self.tw + (x // 16)
        gi = torch.from_numpy(self.flat[self.offs[t]:self.offs[t + 1]])
        dx, dy = self.m2_64[gi, 0] - (x + 0.5), self.m2_64[gi, 1] - (y + 0.5)
        sigma = 0.5 * (self.cn_64[gi, 0] * dx * dx + self.cn_64[gi, 2] * dy * dy) + self.cn_64[gi, 1] * dx * dy
        alpha = torch.clamp(self.op_64[gi] * torch.exp(-sigma), max=0.999)
        valid = (sigma >= 0) & (alpha >= 1.0 / 255.0)
        T = torch.cumprod(1.0 - torch.where(valid, alpha, torch.zeros_like(alpha)), 0)
        alive = torch.cat([torch.ones(1, dtype=torch.bool), T[:-1] > 1e-4])  # entries reached before the stop
        near_skip = ((alpha - 1.0 / 255.0).abs() < 2e-6) & (sigma >= -1e-9) & alive
        near_stop = ((T - 1e-4).abs() < 2e-7) & alive
        near_cap = ((self.op_64[gi] * torch.exp(-sigma) - 0.999).abs() < 2e-6) & alive
        return bool(near_skip.any() | near_stop.any() | near_cap.any())

    def check(self, og, grads, max_bad_radii, oracle_loss=None, grad_tol=3e-3, max_outlier_frac=1e-3, loss_key=None):
        """``og`` / ``grads``: a HIP route's outputs and parameter gradients of the loss whose oracle-side statement is
        ``oracle_loss(out, to)`` (default: ``loss_of``).  The oracle's backward of a loss — the slow part: tens of seconds
        at 6-10 M Gaussians — is run ONCE per loss and kept: the default loss, or any loss the caller names (``loss_key``:
        the same key must mean the same loss), so that several HIP routes / walks are checked against one oracle frame."""
        orf, pr = self.orf, self.pr
        W, H = self.cam.width, self.cam.height
        key = loss_key if loss_key is not None else ("default" if oracle_loss is None else None)
        kept = self._oracle_grads.get(key) if key is not None else None
        if kept is None:
            for v in pr.values():
                v.grad = None
            l_ref = (oracle_loss or self.loss_of)(orf, lambda t: t)
            l_ref.backward(retain_graph=True)
            kept = (l_ref.detach(), {k: v.grad.clone() for k, v in pr.items()})
            if key is not None:
                self._oracle_grads[key] = kept
        l_ref, ref_grads = kept
        # ---- integer outputs
        hip_radii = og["radii"] if "radii" in og else og["info"]["radii"][0]
        bad_r = hip_radii.cpu().reshape(-1) != orf["radii"].reshape(-1)
        n_bad_r = int(bad_r.sum())
        assert n_bad_r == int((bad_r & self.edge_radius).sum()), \
            f"{int((bad_r & ~self.edge_radius).sum())} of {n_bad_r} radii mismatches are not fp32 rounding edges"
        assert n_bad_r <= max_bad_radii  # (context: ~1e-4 of the Gaussians is what an fp32 projection leaves on an edge)
        bad_t = og["info"]["tiles_per_gauss"].cpu()[0] != orf["info"]["tiles_per_gauss"][0]
        assert int(bad_t.sum()) == int((bad_t & self.edge_tiles).sum()), \
            f"{int((bad_t & ~self.edge_tiles).sum())} of {int(bad_t.sum())} tiles_per_gauss mismatches are not fp32 rounding edges"
        # ---- images: 1e-4 of the tensor maximum — and EVERY outlier pixel must sit on a threshold decision of its walk:
        # in fp64, some entry of its tile's list has alpha within 2e-6 of 1/255 (the skip), or the running transmittance
        # passes within 2e-7 of 1e-4 (the stop) — where one Gaussian more or less moves a channel by at most
        # alpha T c <= 1/255
        outliers = torch.zeros(H, W, dtype=torch.bool)
        for k, tol in (("rgb", 1e-4), ("accumulation", 1e-4), ("depth", 1e-3)):
            d = (og[k].detach().cpu() - orf[k].detach()).abs() / max(1.0, float(orf[k].detach().abs().max()))
            outliers |= (d > tol).any(-1)
            assert float(d.max()) <= 1.0 / 255.0 + tol, (k, float(d.max()))
        # (a Gaussian whose radius is an fp32 edge case is in or out of a tile's list: its pixels are explained too)
        unexplained = sum(0 if self._on_a_threshold(y, x) else 1 for y, x in torch.nonzero(outliers).tolist())
        n_out = int(outliers.sum())
        assert unexplained == 0, f"{unexplained} of {n_out} outlier pixels are not on an alpha >= 1/255 / T <= 1e-4 decision"
        assert n_out < max_outlier_frac * H * W, n_out  # (context only: the justification above is the check)
        dn = (og["normal"].detach().cpu() - orf["normal"].detach()).abs()
        assert dn.mean().item() < 1e-5 and (dn > 1e-2).float().mean().item() < 1e-3
        for k in pr:
            e = rel_err(grads[k], ref_grads[k])
            assert e < grad_tol, (k, e)  # 3e-3 of the tensor's own largest gradient
        return dict(n_bad_radii=n_bad_r, n_bad_tiles=int(bad_t.sum()), n_outliers=n_out, oracle_loss=float(l_ref.detach()),
                    visible=int((orf["radii"] > 0).sum()), n_isects=len(self.flat))


def fused_node_route(dev, params):
    """``route(cam, loss_of) -> (outputs, grads)``: the fused autograd node on clones of ``params``."""
    from fusionsense_amd.fused import render_fusionsense_fused

    def route(cam, loss_of):
        pg = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        og = render_fusionsense_fused(pg, cam, sh_degree=3, device=dev)
        loss_of(og, lambda t: t.to(dev)).backward()
        return og, {k: v.grad for k, v in pg.items()}
    return route


_WINDOW = {}  # the most recent shared oracle frame: {key: OracleWindow} (one entry: a frame of 6-10 M Gaussians is large)


def shared_oracle_window(key, make):
    """The oracle frame ``key`` — made by ``make()`` on first use and kept until another key is asked for — so that the
    parametrisations of one test (forward walks, routes) are checked against ONE oracle run instead of one each (round 6:
    the oracle runs were 330 of the GPU suite's 530 s).  The frame depends on scene and camera only, never on the walk."""
    win = _WINDOW.get(key)
    if win is None:
        _WINDOW.clear()
        win = _WINDOW[key] = make()
    return win


def check_fused_node_against_oracle_on_crop(dev, params, cam, max_bad_radii, share_key=None):
    """The fused autograd node against the CPU oracle on ALL of ``params``' Gaussians through ``cam`` (OracleWindow);
    ``share_key``: name of the (scene, camera) frame when several parametrisations check against it."""
    make = lambda: OracleWindow({k: v.detach().cpu() for k, v in params.items()}, cam)  # noqa: E731
    win = shared_oracle_window(share_key, make) if share_key is not None else make()
    og, grads = fused_node_route(dev, params)(cam, win.loss_of)
    return win.check(og, grads, max_bad_radii)

def run_bench_ranks(n_ranks, bench_args, env_extra=None, timeout=240):
    """bench.py under torch.distributed.run the way the driver launches it (one process per rank, 127.0.0.1
    rendezvous), with one retry on a fresh port; returns (parsed JSON line, stderr).  Kills only its own launcher's
    process group on a hang."""
    import json, os, signal, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FSGS_BENCH_VERBOSE="1", **(env_extra or {}))
    out = err = ""
    p = None
    for attempt in range(2):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus",
               str(n_ranks)] + list(bench_args)
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=root,
                             start_new_session=True)
        try:
            out, err = p.communicate(timeout=timeout)
            break
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)  # this launcher's own process group only
            out, err = p.communicate()
            assert attempt == 0, f"{n_ranks}-rank run hung twice:\n" + err[-3000:]
    assert p.returncode == 0, err[-3000:]
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1]), err


def write_scene_dir(root, rng, variant="plain", H=12, W=16, n_pts=60, n_hull=25, look_at=False):
    """A miniature FusionSense scene directory, deterministic in ``rng``: transforms.json (+ images, masks, 16-bit
    depth, mono normals), seed and hull PLYs, gelsight_transform.json with two gel patches.  Shared by the CPU tests,
    the GPU scene-directory test and tests/golden/make_reference_dataparser_goldens.py (which runs the REFERENCE's
    dataparser over the same directory).

    variant "plain":  zero-padded names listed out of order, global intrinsics, .npy touch masks, 3-channel touch normals
    variant "ragged": names that sort differently by string and by number (frame_2 / frame_10 / ...), per-frame
                      intrinsics, applied_transform + applied_scale, a .pcd touch mask and 2-channel touch normals"""
    import json
    from pathlib import Path
    import numpy as np
    from PIL import Image
    from fusionsense_amd import io as fio
    root = Path(root)
    for d in ("images", "masks", "depth", "normals_from_pretrain", "tactile"):
        (root / d).mkdir(parents=True, exist_ok=True)
    ragged = variant == "ragged"
    names = ["frame_10", "frame_2", "frame_1", "frame_21", "frame_3"] if ragged else ["frame_00002", "frame_00000", "frame_00001"]
    frames, c2ws = [], {}
    for i, n in enumerate(names):
        img = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
        Image.fromarray(img).save(root / "images" / f"{n}.png")
        Image.fromarray(((rng.random((H, W)) > 0.4) * 255).astype(np.uint8)).save(root / "masks" / f"{n}.png")
        Image.fromarray(rng.integers(300, 2000, size=(H, W)).astype(np.uint16)).save(root / "depth" / f"{n}.png")
        Image.fromarray(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).save(root / "normals_from_pretrain" / f"{n}.png")
        q = rng.normal(size=(3, 3)); R, _ = np.linalg.qr(q)
        if np.linalg.det(R) < 0:
            R[:, 0] *= -1
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.normal(size=3) * (i + 1)
        if look_at:  # cameras on a ring that SEE the seed points (the file holds OpenCV axes: y / z of the OpenGL pose flipped)
            import math, torch
            from fusionsense_amd import scenes
            az = 2 * math.pi * i / len(names)
            c2w = scenes.look_at_c2w(torch.tensor([2.0 * math.cos(az), 2.0 * math.sin(az), 0.8]), torch.zeros(3)).numpy()
            T = np.eye(4); T[:3, :4] = c2w; T[:3, 1:3] *= -1
        c2ws[n] = T
        fr = {"file_path": f"images/{n}.png", "mask_path": f"masks/{n}.png", "depth_file_path": f"depth/{n}.png",
              "transform_matrix": T.tolist()}
        if ragged:
            fr.update(fl_x=20.0 + i, fl_y=21.0 + 0.5 * i, cx=W / 2 + 0.25 * i, cy=H / 2 - 0.25 * i, w=W, h=H)
        frames.append(fr)
    pts = rng.normal(size=(n_pts, 3)).astype(np.float32) * 0.3
    col = rng.integers(0, 256, size=(n_pts, 3)).astype(np.uint8)
    fio.write_ply(str(root / "merged_pcd.ply"), {"x": pts[:, 0], "y": pts[:, 1], "z": pts[:, 2], "red": col[:, 0],
                                                  "green": col[:, 1], "blue": col[:, 2]})
    hull = rng.normal(size=(n_hull, 3)).astype(np.float32) * 0.1
    fio.write_ply(str(root / "foreground_pcd.ply"), {"x": hull[:, 0], "y": hull[:, 1], "z": hull[:, 2]})
    meta = {"frames": frames, "ply_file_path": "merged_pcd.ply", "object_pc_path": "foreground_pcd.ply"}
    if ragged:
        meta.update(applied_transform=[[0, 1, 0, 0.1], [1, 0, 0, -0.2], [0, 0, -1, 0.3]], applied_scale=0.75)
    else:
        meta.update(fl_x=20.0, fl_y=21.0, cx=8.0 * W / 16, cy=6.0 * H / 12, w=W, h=H)
    (root / "transforms.json").write_text(json.dumps(meta))
    tframes, raws = [], []

    def write_pcd(path, xyz):
        with open(path, "w") as f:
            f.write("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
                    f"WIDTH {len(xyz)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(xyz)}\nDATA ascii\n")
            for r in xyz:
                f.write(f"{r[0]:.6f} {r[1]:.6f} {r[2]:.6f}\n")

    for k in range(2):
        gx, gy = np.meshgrid(np.arange(20), np.arange(15), indexing="ij")
        raw = np.stack([gx.ravel(), gy.ravel(), -rng.random(300) * 40], -1).astype(np.float32)
        write_pcd(root / "tactile" / f"patch_{k}.pcd", raw)
        mask = rng.random(300) > 0.3
        if ragged and k == 1:  # a mask stored as a point cloud whose z is 1 on the contact pixels
            mxyz = raw.copy(); mxyz[:, 2] = mask.astype(np.float32)
            write_pcd(root / "tactile" / f"mask_{k}.pcd", mxyz)
            mask_path = f"tactile/mask_{k}.pcd"
        else:
            np.save(root / "tactile" / f"mask_{k}.npy", mask)
            mask_path = f"tactile/mask_{k}.npy"
        nrm = rng.normal(size=(300, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
        if ragged and k == 0:  # x / y only: z = -sqrt(1 - x^2 - y^2)
            nrm[:, 2] = -np.abs(nrm[:, 2])
            np.save(root / "tactile" / f"normal_{k}.npy", nrm[:, :2].copy())
        else:
            np.save(root / "tactile" / f"normal_{k}.npy", nrm)
        q = rng.normal(size=(3, 3)); R, _ = np.linalg.qr(q)
        T = np.eye(4); T[:3, :3] = R; T[:3, 3] = rng.normal(size=3) * 0.2
        tframes.append({"patch_path": f"tactile/patch_{k}.pcd", "mask_path": mask_path,
                        "normal_path": f"tactile/normal_{k}.npy", "transform_matrix": T.tolist()})
        raws.append((raw, mask, nrm, T))
    (root / "gelsight_transform.json").write_text(json.dumps({"frames": tframes}))
    return names, c2ws, pts, col, hull, raws
