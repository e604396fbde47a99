#!/usr/bin/env python3
"""Mint regression vectors from the fp64 oracle (seeded; SURVEY.md §8c list).  These pin the
ORACLE against accidental edits — they are self-minted, not reference outputs (the reference
ships none for this path: parity unpinned).  Writes tests/golden/oracle_vectors.npz."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fusionsense_amd import scenes  # noqa: E402
from oracle import gsplat_ref as R  # noqa: E402
from tests.helpers import activated, adversarial_gaussians, camera_mats  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")


def main():
    out = {}
    params, cam = scenes.cube_scene(1000, seed=0)
    viewmat, K = camera_mats(cam)
    W, H = cam.width, cam.height
    # (1) projection of the adversarial set (fp64)
    means, quats, scl, opac = adversarial_gaussians(cam)
    r = R.project(means.double(), quats.double(), scl.double(), viewmat.double(), K.double(), W, H)
    out["proj_radii"], out["proj_means2d"], out["proj_depths"], out["proj_conics"], out["proj_comp"] = (
        r[0].numpy(), r[1].numpy(), r[2].numpy(), r[3].numpy(), r[4].numpy())
    # (2) binning of the same set, both rules (fp32 inputs as the kernels see them)
    r32 = R.project(means, quats, scl, viewmat, K, W, H)
    for legacy in (False, True):
        tpg, ids, flat = R.isect_tiles(r32[1], r32[0], r32[2], 16, 8, 8, legacy=legacy)
        ids_s, flat_s = R.sort_isects(ids, flat)
        tag = "legacy" if legacy else "new"
        out[f"isect_{tag}_tpg"], out[f"isect_{tag}_ids_sorted"], out[f"isect_{tag}_flat_sorted"] = tpg, ids_s, flat_s
        out[f"isect_{tag}_offsets"] = R.isect_offset_encode(ids_s, 1, 8, 8)
    # (6) SH at the axes and one random direction, degrees 0..4
    dirs = torch.tensor([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1],
                         [0.3, -0.5, 0.81]], dtype=torch.float64)
    out["sh_dirs"] = dirs.numpy()
    for d in range(5):
        out[f"sh_bases_deg{d}"] = R.sh_bases(d, torch.nn.functional.normalize(dirs, dim=-1)).numpy()
    # (8) config #1 full frame, fp64: per-channel means + one 16x16 patch + integer meta checksums
    act = {k: v.double() for k, v in activated(params).items()}
    render, alpha, meta = R.rasterization(act["means"], act["quats"], act["scales"], act["opacities"],
                                          act["colors"], viewmat.double(), K.double(), W, H, sh_degree=3,
                                          packed=False, render_mode="RGB+ED")
    out["cfg1_render_mean"] = render.mean(dim=(0, 1, 2)).numpy()
    out["cfg1_alpha_mean"] = alpha.mean().numpy()
    out["cfg1_patch"] = render[0, 56:72, 56:72].numpy()
    out["cfg1_n_isects"] = np.array(meta["flatten_ids"].numel())
    out["cfg1_flat_sum"] = np.array(int(meta["flatten_ids"].long().sum()))
    out["cfg1_offsets"] = meta["isect_offsets"].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
