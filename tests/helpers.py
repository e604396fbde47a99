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


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


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
