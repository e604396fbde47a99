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
