"""On-disk formats in front of the hot path: a FusionSense scene directory -> cameras, seed
Gaussians, per-view supervision and touch patches, host-side only.

Mirrors ``NormalNerfstudio._generate_dataparser_outputs`` (/root/reference/dn_splatter/data/normal_nerfstudio.py:136-725)
and ``GDataset.get_metadata`` (dn_splatter/data/dn_dataset.py:110-243) for the configuration FusionSense runs
(``scripts/train.py:105-146``: ``normal-nerfstudio --load-pcd-normals True --load-3D-points True --normal-format opencv``,
orientation_method = center_method = "none", auto_scale_poses = True, normal_nerfstudio.py:74-77):

  transforms.json      intrinsics (global or per frame), frames sorted by file name, ``transform_matrix`` per frame;
                       camera convention: the y and z camera axes are flipped on load (``poses[:, :3, 1:3] *= -1``,
                       :314), no re-orientation / re-centring, translations scaled so that max |t| = 1 (:320-327)
  ply_file_path        seed points (merged_pcd.ply) and ``object_pc_path`` (the visual hull), both taken from
                       ``output_dir`` and moved by the same transform + scale (:497-540)
  masks / depths / normals_from_pretrain/*.png   per-view supervision files
  gelsight_transform.json + tactile/...          touch patches (:593-685)

nerfstudio's own helpers the reference inherits (``_load_3D_points``, ``get_depth_image_from_path``, natsort, the image
/ mask loading of ``InputDataset``) are not vendored: restated from nerfstudio 1.1.3 as recalled.
open3d's PCA normal estimation (:97-106) is restated with a k-d tree + eigen-decomposition; its normals' SIGNS are
arbitrary there too, and they only seed a disc's orientation, which does not depend on the sign.
"""
from __future__ import annotations

import glob
import json
import os
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch
from torch import Tensor

from .io import read_pcd, read_ply
from .scenes import Camera


@dataclass
class DataparserOutputs:
    cameras: List[Camera]
    image_filenames: List[Path]
    mask_filenames: Optional[List[Path]]
    dataparser_scale: float
    dataparser_transform: Tensor  # [3,4]
    metadata: Dict = field(default_factory=dict)


def natural_key(path) -> list:
    """natsort's default ordering for file names: digit runs compare as numbers."""
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", str(path))]


def mut_and_scale(points3D: Tensor, transform_matrix: Tensor, scale_factor: float) -> Tensor:
    """normal_nerfstudio.py:37-49: homogeneous transform by a [3,4] matrix, then a uniform scale."""
    points3D = torch.cat((points3D, torch.ones_like(points3D[..., :1])), -1) @ transform_matrix.T
    return points3D * scale_factor


def auto_orient_and_center_poses(poses: Tensor, method: str = "none", center_method: str = "none"):
    """nerfstudio camera_utils.auto_orient_and_center_poses for the methods FusionSense uses (``none`` / ``none``,
    normal_nerfstudio.py:74-75) plus centring on the mean camera position.  Returns (poses [N,3,4], transform [3,4])."""
    origins = poses[..., :3, 3]
    if center_method == "poses":
        translation = torch.mean(origins, dim=0)
    elif center_method == "none":
        translation = torch.zeros_like(origins[0])
    else:
        raise NotImplementedError(f"center_method {center_method!r} (FusionSense uses 'none')")
    if method != "none":
        raise NotImplementedError(f"orientation method {method!r} (FusionSense uses 'none')")
    transform = torch.eye(4)
    transform[:3, 3] = -translation
    transform = transform[:3, :]
    return transform @ _to4x4(poses), transform


def _to4x4(poses: Tensor) -> Tensor:
    if poses.shape[-2] == 4:
        return poses
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).expand(poses.shape[0], 1, 4)
    return torch.cat([poses, bottom], dim=1)


def estimate_normals(points: Tensor, radius: float = 0.1, max_nn: int = 30) -> Tensor:
    """open3d ``estimate_normals(KDTreeSearchParamHybrid(radius, max_nn))`` + ``normalize_normals``: the eigenvector
    of the smallest eigenvalue of the covariance of up to ``max_nn`` neighbours within ``radius``; (0,0,1) where
    fewer than 3 neighbours exist.  Sign as the eigen-solver leaves it (it is arbitrary in open3d as well)."""
    from sklearn.neighbors import NearestNeighbors
    x = points.detach().cpu().numpy().astype(np.float64)
    n = len(x)
    k = min(max_nn, n)
    dist, idx = NearestNeighbors(n_neighbors=k, algorithm="auto").fit(x).kneighbors(x)
    ok = dist <= radius                                   # [n,k]
    cnt = ok.sum(1)
    nb = x[idx]                                           # [n,k,3]
    w = ok[..., None].astype(np.float64)
    mean = (nb * w).sum(1) / np.maximum(cnt, 1)[:, None]
    d = (nb - mean[:, None, :]) * w
    cov = np.einsum("nki,nkj->nij", d, d) / np.maximum(cnt, 1)[:, None, None]
    _, vec = np.linalg.eigh(cov)
    normals = vec[:, :, 0]
    normals[cnt < 3] = np.array([0.0, 0.0, 1.0])
    normals /= np.maximum(np.linalg.norm(normals, axis=-1, keepdims=True), 1e-30)
    return torch.from_numpy(normals.astype(np.float32))


def load_3d_points(ply_file_path, transform_matrix: Tensor, scale_factor: float) -> Optional[Dict[str, Tensor]]:
    """nerfstudio ``Nerfstudio._load_3D_points``: xyz moved into dataparser coordinates, colours as uint8."""
    p = read_ply(str(ply_file_path))
    if len(p["x"]) == 0:
        return None
    pts = torch.from_numpy(np.stack([p["x"], p["y"], p["z"]], -1).astype(np.float32))
    pts = mut_and_scale(pts, transform_matrix, scale_factor)
    if all(k in p for k in ("red", "green", "blue")):
        col = np.stack([p["red"], p["green"], p["blue"]], -1)
        if np.issubdtype(col.dtype, np.floating):
            col = col * 255.0  # (open3d holds colours in [0,1]; the dataparser multiplies by 255)
        rgb = torch.from_numpy(col.astype(np.uint8))
    else:
        rgb = torch.zeros(pts.shape[0], 3, dtype=torch.uint8)
    return {"points3D_xyz": pts, "points3D_rgb": rgb}


def load_touch_patches(data: Path, transform_matrix: Tensor, scale_factor: float, gel_scale_factor: float = 6.34e-5,
                       touch_downsample_factor: int = 5) -> List[Dict[str, Tensor]]:
    """normal_nerfstudio.py:593-685: every frame of gelsight_transform.json -> {points_xyz, points_rgb, normals, bbox}.
    The gel point cloud holds integer pixel indices in x / y: centred, scaled by the gel pitch, moved by the frame's
    transform and then by the dataparser's; the mask picks the contact pixels; normals are rotated and translated by
    the frame's [3,4] transform exactly as the reference does (``mut_and_scale(pcd_normal3D, tr[:3, :], 1.0)``);
    the oriented box spans the raw patch in x / y and 5x its depth range behind the contact plane."""
    data = Path(data)
    with open(data / "gelsight_transform.json", "r", encoding="utf-8") as f:
        touch_meta = json.load(f)
    patches = []
    for touchframe in touch_meta["frames"]:
        raw = read_pcd(str(data / touchframe["patch_path"]))
        raw_pcd = torch.from_numpy(np.stack([raw["x"], raw["y"], raw["z"]], -1).astype(np.float32))
        raw_pcd = raw_pcd[::touch_downsample_factor, :]
        tr = torch.tensor(touchframe["transform_matrix"], dtype=raw_pcd.dtype)
        pcd = raw_pcd.clone()
        pcd[:, :2] -= torch.mean(raw_pcd, dim=0)[:2]
        pcd *= gel_scale_factor
        pcd = mut_and_scale(pcd, tr[:3, :], 1.0)
        pcd = mut_and_scale(pcd, transform_matrix, scale_factor)
        mask_path = str(touchframe["mask_path"])
        if mask_path.endswith(".pcd"):
            mask = np.asarray(read_pcd(str(data / mask_path))["z"]) == 1
        elif mask_path.endswith(".npy"):
            mask = np.load(str(data / mask_path))
        else:
            raise KeyError("Unsupported mask type")
        mask = np.asarray(mask).reshape(-1)[::touch_downsample_factor].astype(bool)
        np_pts = pcd[torch.from_numpy(mask)]
        nrm = torch.from_numpy(np.load(str(data / touchframe["normal_path"])))
        nrm = nrm[::touch_downsample_factor, :] if nrm.dim() == 2 else nrm.reshape(-1, nrm.shape[-1])[::touch_downsample_factor]
        if nrm.shape[-1] == 2:
            nrm = nrm.reshape(-1, 2)[torch.from_numpy(mask)]
            x, y = nrm[..., 0], nrm[..., 1]
            z = -torch.sqrt(torch.clamp(1.0 - x ** 2 - y ** 2, min=0.0))
        elif nrm.shape[-1] == 3:
            nrm = nrm.reshape(-1, 3)[torch.from_numpy(mask)]
            x, y, z = nrm[..., 0], nrm[..., 1], nrm[..., 2]
        else:
            raise KeyError("Unsupported Normal Type")
        normal3d = torch.stack((x, y, z)).to(dtype=torch.float32).T
        normal3d = mut_and_scale(normal3d, tr[:3, :], 1.0)
        x_diff = torch.abs(torch.max(raw_pcd[:, 0]) - torch.min(raw_pcd[:, 0]))
        y_diff = torch.abs(torch.max(raw_pcd[:, 1]) - torch.min(raw_pcd[:, 1]))
        z_diff = torch.abs(torch.max(raw_pcd[:, 2]) - torch.min(raw_pcd[:, 2]))
        lo = [-x_diff / 2, -y_diff / 2, -z_diff * 5]
        hi = [x_diff / 2, y_diff / 2, 0]
        aabb = torch.tensor([[(hi if (i >> 2) & 1 else lo)[0], (hi if (i >> 1) & 1 else lo)[1], (hi if i & 1 else lo)[2]]
                             for i in range(8)], dtype=pcd.dtype)
        aabb = aabb * gel_scale_factor
        aabb = mut_and_scale(aabb, tr[:3, :], 1.0)
        aabb = mut_and_scale(aabb, transform_matrix, scale_factor)
        patches.append({"points_xyz": np_pts, "points_rgb": torch.zeros_like(np_pts), "normals": normal3d, "bbox": aabb})
    return patches


def parse_transforms(data, output_dir=None, split: str = "train", auto_scale_poses: bool = True,
                     scale_factor: float = 1.0, orientation_method: str = "none", center_method: str = "none",
                     load_3D_points: bool = True, load_pcd_normals: bool = True, load_normals: bool = True,
                     normal_format: str = "opencv", load_touches: bool = False, gel_scale_factor: float = 6.34e-5,
                     depth_unit_scale_factor: float = 1e-3) -> DataparserOutputs:
    """``NormalNerfstudio._generate_dataparser_outputs`` (normal_nerfstudio.py:136-725)."""
    data = Path(data)
    output_dir = Path(output_dir) if output_dir is not None else data
    with open(data / "transforms.json", "r", encoding="utf-8") as f:
        meta = json.load(f)
    fixed = {k: (k in meta) for k in ("fl_x", "fl_y", "cx", "cy", "h", "w")}
    frames = [meta["frames"][i] for i in np.argsort([str(data / Path(fr["file_path"])) for fr in meta["frames"]])]
    image_filenames, mask_filenames, depth_filenames, poses = [], [], [], []
    per = {k: [] for k in fixed}
    for frame in frames:
        for k, is_fixed in fixed.items():
            if not is_fixed:
                assert k in frame, f"{k} not specified in frame"
                per[k].append(float(frame[k]))
        image_filenames.append(data / Path(frame["file_path"]))
        poses.append(np.array(frame["transform_matrix"]))
        if "mask_path" in frame:
            mask_filenames.append(output_dir / Path(frame["mask_path"]))  # (masks come from output_dir, :222-228)
        if "depth_file_path" in frame:
            depth_filenames.append(data / Path(frame["depth_file_path"]))
    assert len(mask_filenames) in (0, len(image_filenames)), "mask_path must be given for every frame or for none"
    assert len(depth_filenames) in (0, len(image_filenames)), "depth_file_path must be given for every frame or none"
    normal_filenames = sorted(glob.glob(f"{output_dir / 'normals_from_pretrain'}/*.png"), key=natural_key)
    # natural order of the image names (:253-259), then — unless the split names its files — the same permutation is
    # applied once more further down (:331-346): reproduced as written (identity for zero-padded names)
    order = sorted(range(len(image_filenames)), key=lambda i: natural_key(image_filenames[i]))
    image_filenames = [image_filenames[i] for i in order]
    depth_filenames = [depth_filenames[i] for i in order] if depth_filenames else []
    poses = [poses[i] for i in order]
    mask_filenames = [mask_filenames[i] for i in order] if mask_filenames else []
    indices = list(order)  # (per-frame intrinsics stay in file-name order and are indexed by ``indices``, :384-414)
    if f"{split}_filenames" in meta:
        wanted = {data / Path(x) for x in meta[f"{split}_filenames"]}
        missing = wanted.difference(image_filenames)
        if missing:
            raise RuntimeError(f"Some filenames for split {split} were not found: {missing}.")
        indices = [i for i, p in enumerate(image_filenames) if p in wanted]
    poses_t = torch.from_numpy(np.array(poses).astype(np.float32))
    poses_t[:, :3, 1:3] *= -1  # FusionSense's files hold OpenCV camera axes; the model works in nerfstudio's (OpenGL)
    poses_t, transform_matrix = auto_orient_and_center_poses(
        poses_t, method=meta.get("orientation_override", orientation_method), center_method=center_method)
    scale = 1.0
    if auto_scale_poses:
        scale /= float(torch.max(torch.abs(poses_t[:, :3, 3])))
    scale *= scale_factor
    poses_t[:, :3, 3] *= scale
    image_filenames = [image_filenames[i] for i in indices]
    mask_filenames = [mask_filenames[i] for i in indices] if mask_filenames else []
    depth_filenames = [depth_filenames[i] for i in indices] if depth_filenames else []
    normal_filenames = [Path(normal_filenames[i]) for i in indices] if normal_filenames else []
    stems = [p.stem for p in image_filenames]
    for name in normal_filenames:
        assert name.stem in stems, name
    poses_t = poses_t[torch.tensor(indices, dtype=torch.long)]

    def intr(k, j, cast=float):
        return cast(meta[k]) if fixed[k] else cast(per[k][indices[j]])

    cameras = [Camera(poses_t[j, :3, :4].contiguous(), intr("fl_x", j), intr("fl_y", j), intr("cx", j), intr("cy", j),
                      intr("w", j, int), intr("h", j, int)) for j in range(len(indices))]
    applied_scale = float(meta["applied_scale"]) if "applied_scale" in meta else 1.0
    scale *= applied_scale
    dataparser_transform = transform_matrix
    if "applied_transform" in meta:
        at = torch.tensor(meta["applied_transform"], dtype=transform_matrix.dtype)
        dataparser_transform = transform_matrix @ torch.cat([at, torch.tensor([[0.0, 0.0, 0.0, 1.0]])], 0)
    metadata: Dict = {}
    if load_3D_points and "ply_file_path" in meta:
        sp = load_3d_points(output_dir / meta["ply_file_path"], transform_matrix, scale)
        if sp is not None:
            metadata.update(sp)
    if "object_pc_path" in meta:
        hull = load_3d_points(output_dir / meta["object_pc_path"], transform_matrix, scale)
        if hull is not None:
            metadata["visual_hull"] = hull["points3D_xyz"]
    if load_pcd_normals and "points3D_xyz" in metadata:
        nrm = estimate_normals(metadata["points3D_xyz"])
        # (:107-113: the normals go through the same [3,4] transform, translation included, as written)
        metadata["points3D_normals"] = torch.cat((nrm, torch.ones_like(nrm[..., :1])), -1) @ transform_matrix.T
    if load_normals:
        metadata.update(normal_filenames=normal_filenames, load_normals=True, normal_format=normal_format)
    if load_touches:
        metadata["load_touches"] = True
        metadata["touch_patches"] = load_touch_patches(data, transform_matrix, scale, gel_scale_factor)
        metadata["gel_scale_factor"] = gel_scale_factor
    metadata.update(scale_factor=scale, transform_matrix=transform_matrix,
                    depth_filenames=depth_filenames if depth_filenames else None,
                    depth_unit_scale_factor=depth_unit_scale_factor)
    return DataparserOutputs(cameras=cameras, image_filenames=image_filenames,
                             mask_filenames=mask_filenames if mask_filenames else None, dataparser_scale=scale,
                             dataparser_transform=dataparser_transform, metadata=metadata)


def _open_image(path) -> np.ndarray:
    from PIL import Image
    return np.array(Image.open(str(path)))


def get_depth_image_from_path(filepath, height: int, width: int, scale_factor: float) -> Tensor:
    """nerfstudio data_utils.get_depth_image_from_path: .npy or 16-bit PNG -> [H,W,1] in scene units, nearest resize."""
    filepath = Path(filepath)
    image = np.load(filepath) if filepath.suffix == ".npy" else _open_image(filepath)
    image = image.astype(np.float64) * scale_factor
    if image.shape[:2] != (height, width):
        ys = (np.arange(height) * image.shape[0] / height).astype(np.int64)
        xs = (np.arange(width) * image.shape[1] / width).astype(np.int64)
        image = image[ys][:, xs]
    return torch.from_numpy(image[:, :, None])


def get_normal_image_from_path(path, normal_format: str = "opencv") -> Tensor:
    """GDataset.get_normal_image_from_path (dn_dataset.py:182-215), camera frame: PNG -> [H,W,3] in [0,1];
    ``opengl`` files are flipped to OpenCV axes and mapped back to [0,1]."""
    path = Path(path)
    if path.suffix == ".png":
        normal_map = _open_image(path).astype("uint8")[..., :3]
    else:
        normal_map = np.load(path).transpose(1, 2, 0)
        if normal_map.min() < 0:
            normal_map = (normal_map + 1) / 2
    normal_map = torch.from_numpy(normal_map.astype("float32") / 255.0).float()
    if normal_format == "opengl":
        h, w, _ = normal_map.shape
        normal_map = (2 * normal_map.view(-1, 3) - 1) @ torch.diag(torch.tensor([1.0, -1.0, -1.0]))
        normal_map = normal_map.view(h, w, 3)
        if normal_map.min() < 0:
            normal_map = (normal_map + 1) / 2
    return normal_map


def load_batch(outputs: DataparserOutputs, idx: int) -> Dict[str, Tensor]:
    """One view's batch as the reference's datamanager hands it to get_loss_dict: ``image`` float [H,W,3] in [0,1]
    (InputDataset.get_image_float32), ``mask`` bool [H,W,1], ``sensor_depth`` [H,W,1] in scene units
    (depth_unit_scale_factor * dataparser_scale, dn_dataset.py:119-128), ``normal`` [H,W,3]."""
    cam = outputs.cameras[idx]
    img = _open_image(outputs.image_filenames[idx])
    if img.ndim == 2:
        img = np.repeat(img[..., None], 3, axis=2)
    batch = {"image": torch.from_numpy(img.astype("float32") / 255.0)[..., :3].contiguous(), "image_idx": idx}
    if outputs.mask_filenames is not None:
        m = _open_image(outputs.mask_filenames[idx])
        m = m[..., 0] if m.ndim == 3 else m
        batch["mask"] = torch.from_numpy(m)[..., None].bool()
    md = outputs.metadata
    if md.get("depth_filenames"):
        scale = md["depth_unit_scale_factor"] * outputs.dataparser_scale
        batch["sensor_depth"] = get_depth_image_from_path(md["depth_filenames"][idx], cam.height, cam.width, scale).float()
    if md.get("load_normals") and md.get("normal_filenames"):
        batch["normal"] = get_normal_image_from_path(md["normal_filenames"][idx], md.get("normal_format", "opencv"))
    return batch


def seed_gauss_params(outputs: DataparserOutputs, sh_degree: int = 3) -> Dict[str, Tensor]:
    """populate_modules (dn_model.py:196-300) from the dataparser's seed points (+ their PCA normals)."""
    from .io import init_gauss_params
    md = outputs.metadata
    return init_gauss_params(md["points3D_xyz"], md["points3D_rgb"].float(), md.get("points3D_normals"), sh_degree)
