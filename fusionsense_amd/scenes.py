"""Synthetic scenes for the BASELINE.json configurations.

There is no network for datasets or checkpoints, so every scene is seeded random data of
the shape the reference trains on.  Parameters are returned in the *stored* (pre-activation)
form FusionSense keeps in ``gauss_params`` (dn_splatter/dn_model.py:294-304): log-scales,
opacity logits, SH split into ``features_dc`` [N,3] and ``features_rest`` [N,K-1,3].
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
from torch import Tensor

C0 = 0.28209479177387814


def rgb_to_sh(rgb: Tensor) -> Tensor:
    """nerfstudio RGB2SH (imported at dn_model.py:47): (rgb - 0.5) / C0."""
    return (rgb - 0.5) / C0


def sh_to_rgb(sh: Tensor) -> Tensor:
    """dn_model.py:2050-2055."""
    return sh * C0 + 0.5


def random_quat_tensor(n: int, generator=None) -> Tensor:
    """Uniform random unit quaternions, same construction as dn_model.py:2035-2047."""
    u = torch.rand(n, generator=generator)
    v = torch.rand(n, generator=generator)
    w = torch.rand(n, generator=generator)
    return torch.stack(
        [
            torch.sqrt(1 - u) * torch.sin(2 * math.pi * v),
            torch.sqrt(1 - u) * torch.cos(2 * math.pi * v),
            torch.sqrt(u) * torch.sin(2 * math.pi * w),
            torch.sqrt(u) * torch.cos(2 * math.pi * w),
        ],
        dim=-1,
    )


def look_at_c2w(eye: Tensor, target: Tensor, up=(0.0, 0.0, 1.0)) -> Tensor:
    """OpenGL-convention camera-to-world [3,4]: camera looks along -z, +y up."""
    eye = eye.to(torch.float32)
    fwd = torch.nn.functional.normalize(target.to(torch.float32) - eye, dim=0)
    upv = torch.tensor(up, dtype=torch.float32)
    right = torch.linalg.cross(fwd, upv)
    if right.norm() < 1e-6:
        right = torch.linalg.cross(fwd, torch.tensor([0.0, 1.0, 0.0]))
    right = torch.nn.functional.normalize(right, dim=0)
    true_up = torch.linalg.cross(right, fwd)
    return torch.stack([right, true_up, -fwd, eye], dim=1)


@dataclass
class Camera:
    c2w: Tensor  # [3,4] OpenGL
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int

    def K(self) -> Tensor:
        return torch.tensor([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])


def _sh_params(n: int, sh_degree: int, g) -> Tuple[Tensor, Tensor]:
    k = (sh_degree + 1) ** 2
    dc = rgb_to_sh(torch.rand(n, 3, generator=g))
    rest = 0.1 * torch.randn(n, k - 1, 3, generator=g)
    return dc, rest


def cube_scene(n: int = 1000, seed: int = 0, sh_degree: int = 3) -> Tuple[Dict[str, Tensor], Camera]:
    """Config #1: Gaussians on the surface of [-0.5,0.5]^3, one 128x128 camera."""
    g = torch.Generator().manual_seed(seed)
    face = torch.randint(0, 6, (n,), generator=g)
    uv = torch.rand(n, 2, generator=g) - 0.5
    axis = face // 2
    sign = (face % 2).to(torch.float32) - 0.5
    means = torch.zeros(n, 3)
    for a in range(3):
        sel = axis == a
        others = [i for i in range(3) if i != a]
        means[sel, a] = sign[sel]
        means[sel, others[0]] = uv[sel, 0]
        means[sel, others[1]] = uv[sel, 1]
    lo, hi = math.log(0.01), math.log(0.05)
    scales = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    quats = random_quat_tensor(n, g)
    opac = torch.randn(n, 1, generator=g)
    dc, rest = _sh_params(n, sh_degree, g)
    params = dict(means=means, scales=scales, quats=quats, features_dc=dc, features_rest=rest, opacities=opac)
    # looking at the origin from (0,0,2.5); "up" chosen as +y because the view axis is z
    cam = Camera(look_at_c2w(torch.tensor([0.0, 0.0, 2.5]), torch.zeros(3), up=(0.0, 1.0, 0.0)),
                 128.0, 128.0, 64.0, 64.0, 128, 128)
    return params, cam


def lego_like_scene(
    n: int = 300_000, seed: int = 0, sh_degree: int = 3, n_boxes: int = 200
) -> Dict[str, Tensor]:
    """Config #2 Gaussians: 80 % on the faces of random boxes inside [-1,1]^3, 20 % uniform;
    disc-like scales (one axis x0.1, the `two_d_gaussians` prior dn_model.py:98,817-819);
    opacity logits N(2,1.5)."""
    g = torch.Generator().manual_seed(seed)
    n_surf = int(0.8 * n)
    centers = torch.rand(n_boxes, 3, generator=g) * 1.6 - 0.8
    half = 0.03 + 0.17 * torch.rand(n_boxes, 3, generator=g)
    b = torch.randint(0, n_boxes, (n_surf,), generator=g)
    face = torch.randint(0, 6, (n_surf,), generator=g)
    uvw = torch.rand(n_surf, 3, generator=g) * 2 - 1
    axis = face // 2
    sign = (face % 2).to(torch.float32) * 2 - 1
    uvw[torch.arange(n_surf), axis] = sign
    surf = (centers[b] + half[b] * uvw).clamp(-1, 1)
    vol = torch.rand(n - n_surf, 3, generator=g) * 2 - 1
    means = torch.cat([surf, vol], 0)
    lo, hi = math.log(0.003), math.log(0.03)
    scales = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    thin = torch.randint(0, 3, (n,), generator=g)
    scales[torch.arange(n), thin] += math.log(0.1)
    quats = random_quat_tensor(n, g)
    opac = 2.0 + 1.5 * torch.randn(n, 1, generator=g)
    dc, rest = _sh_params(n, sh_degree, g)
    perm = torch.randperm(n, generator=g)  # storage order carries no spatial locality
    return dict(
        means=means[perm], scales=scales[perm], quats=quats[perm],
        features_dc=dc[perm], features_rest=rest[perm], opacities=opac[perm],
    )


def hemisphere_cameras(
    n_views: int = 100, radius: float = 4.031, width: int = 800, height: int = 800,
    focal: float = 1111.11, seed: int = 0,
) -> list:
    """Blender-synthetic style rig: cameras on the upper hemisphere looking at the origin."""
    g = torch.Generator().manual_seed(seed + 12345)
    cams = []
    for _ in range(n_views):
        az = 2 * math.pi * torch.rand(1, generator=g).item()
        el = math.radians(10.0 + 70.0 * torch.rand(1, generator=g).item())
        eye = radius * torch.tensor([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), math.sin(el)])
        cams.append(Camera(look_at_c2w(eye, torch.zeros(3)), focal, focal, width / 2.0, height / 2.0, width, height))
    return cams


def fusionsense_like_scene(seed: int = 0, n_hull: int = 20_000, n_bg: int = 40_000, sh_degree: int = 3):
    """Config #3: a dense 0.1-radius object blob ('visual hull' seeds, utils/VisualHull.py:135)
    inside a sparse background shell (utils/generate_pcd.py:99), 9 views at 1280x720 with the
    RealSense-like intrinsics the reference preprocesses for
    (utils/metric3dv2_depth_generation.py:55)."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n_hull, 3, generator=g)
    hull = 0.1 * d / d.norm(dim=-1, keepdim=True) * torch.rand(n_hull, 1, generator=g) ** (1 / 3)
    bg = torch.rand(n_bg, 3, generator=g) * 1.6 - 0.8
    bg[:, 2] = bg[:, 2] * 0.25 - 0.3
    means = torch.cat([hull, bg], 0)
    n = means.shape[0]
    scales = torch.cat([
        torch.full((n_hull, 3), math.log(0.005)) + 0.3 * torch.randn(n_hull, 3, generator=g),
        torch.full((n_bg, 3), math.log(0.02)) + 0.3 * torch.randn(n_bg, 3, generator=g),
    ])
    quats = random_quat_tensor(n, g)
    opac = torch.logit(torch.full((n, 1), 0.1)) + 0.5 * torch.randn(n, 1, generator=g)
    dc, rest = _sh_params(n, sh_degree, g)
    params = dict(means=means, scales=scales, quats=quats, features_dc=dc, features_rest=0.0 * rest, opacities=opac)
    cams = []
    for i in range(9):
        az = 2 * math.pi * i / 9
        eye = torch.tensor([math.cos(az), math.sin(az), 0.35])
        eye = eye / eye.abs().max()  # max |translation| = 1 (normal_nerfstudio.py:325-330)
        cams.append(Camera(look_at_c2w(eye, torch.zeros(3)), 641.299, 641.299, 636.707, 362.299, 1280, 720))
    return params, cams


def bicycle_like_scene(n: int = 6_000_000, seed: int = 0, sh_degree: int = 3) -> Dict[str, Tensor]:
    """Config #4/#5 Gaussians: unbounded-scene-like radial distribution, small splats."""
    g = torch.Generator().manual_seed(seed)
    d = torch.randn(n, 3, generator=g)
    r = 0.2 + 3.0 * torch.rand(n, 1, generator=g) ** 2
    means = d / d.norm(dim=-1, keepdim=True) * r
    means[:, 2] = means[:, 2].abs() * 0.3 - 0.2
    lo, hi = math.log(0.002), math.log(0.02)
    scales = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    quats = random_quat_tensor(n, g)
    opac = 1.0 + 1.5 * torch.randn(n, 1, generator=g)
    dc, rest = _sh_params(n, sh_degree, g)
    return dict(means=means, scales=scales, quats=quats, features_dc=dc, features_rest=rest, opacities=opac)


def _quats_from_normals(normals: Tensor, g) -> Tensor:
    """Unit quaternions (wxyz) whose rotation takes the local z axis to ``normals`` [n,3], with a random spin about
    it: the orientation of a disc-like Gaussian lying IN a surface (its thin axis is the surface normal, which is also
    what dn_model.py:618-636 reads back as the Gaussian's normal)."""
    n = torch.nn.functional.normalize(normals, dim=-1)
    q = torch.stack([1.0 + n[:, 2], -n[:, 1], n[:, 0], torch.zeros_like(n[:, 0])], dim=-1)
    flip = q[:, 0] < 1e-6  # normal = -z: any half turn about an in-plane axis
    q[flip] = torch.tensor([0.0, 1.0, 0.0, 0.0])
    q = torch.nn.functional.normalize(q, dim=-1)
    t = math.pi * torch.rand(n.shape[0], generator=g)
    c, s_ = torch.cos(t), torch.sin(t)
    w, x, y, z = q.unbind(-1)
    return torch.stack([w * c - z * s_, x * c + y * s_, y * c - x * s_, w * s_ + z * c], dim=-1)  # q * (c, 0, 0, s)


def surface_dense_scene(n: int = 6_000_000, seed: int = 0, sh_degree: int = 3, extent: float = 6.0,
                        n_objects: int = 1200, scale_lo: float = 0.002, scale_hi: float = 0.015,
                        opaque_frac: float = 0.5, floater_frac: float = 0.08) -> Dict[str, Tensor]:
    """Configs #4 / #5 (round 4): a scene whose Gaussians sit ON SURFACES that cameras see — what a trained capture of
    an unbounded scene looks like to the rasterizer — instead of a fog with the cameras inside it
    (``bicycle_like_scene``, kept as ``--scene volume``):

    * 25 % on a ground disc of radius ``extent`` (z = 0, slight relief), 67 % on the faces of ``n_objects`` boxes of
      half-size 0.05 .. 0.3 standing on the ground inside 0.85 x extent, ``floater_frac`` semi-transparent floaters;
    * config #2's scale law at two thirds of its size (exp U(log ``scale_lo``, log ``scale_hi``), one axis x 0.1: 20 times
      the Gaussians of config #2 describe finer detail), the thin axis ALONG the surface normal (discs lying in their
      surface; floaters oriented at random);
    * bimodal opacities: ``opaque_frac`` of the surface Gaussians logit N(3, 1), the others N(-2.5, 1); floaters N(-3, 1);
    * storage order shuffled (no spatial locality in the rows).
    Rendered from ``ring_cameras`` OUTSIDE the content, every surface faces some camera, a ray meets the objects'
    front faces, then what lies behind them: pixels saturate on the first surfaces and the lists behind them are
    occluded — not empty."""
    g = torch.Generator().manual_seed(seed)
    n_float = int(floater_frac * n)
    n_ground = int(0.25 * n)
    n_obj = n - n_float - n_ground
    # ground disc
    r = extent * torch.sqrt(torch.rand(n_ground, generator=g))
    a = 2 * math.pi * torch.rand(n_ground, generator=g)
    ground = torch.stack([r * torch.cos(a), r * torch.sin(a), 0.004 * torch.randn(n_ground, generator=g)], dim=-1)
    ground_n = torch.tensor([0.0, 0.0, 1.0]).expand(n_ground, 3) + 0.03 * torch.randn(n_ground, 3, generator=g)
    # boxes standing on the ground
    half = 0.05 + 0.25 * torch.rand(n_objects, 3, generator=g)
    rc = 0.85 * extent * torch.sqrt(torch.rand(n_objects, generator=g))
    ac = 2 * math.pi * torch.rand(n_objects, generator=g)
    centers = torch.stack([rc * torch.cos(ac), rc * torch.sin(ac), half[:, 2]], dim=-1)
    yaw = 2 * math.pi * torch.rand(n_objects, generator=g)
    # faces picked by area, so the Gaussian density per unit area is the same on every face of every box
    fa = torch.stack([half[:, 1] * half[:, 2], half[:, 0] * half[:, 2], half[:, 0] * half[:, 1]], dim=-1)  # x, y, z faces
    w_ = (fa / fa.sum()).repeat_interleave(2, dim=1).reshape(-1) / 2  # [n_objects * 6]
    pick = torch.multinomial(w_, n_obj, replacement=True, generator=g)
    b, face = pick // 6, pick % 6
    axis, sign = face // 2, (face % 2).to(torch.float32) * 2 - 1
    uvw = torch.rand(n_obj, 3, generator=g) * 2 - 1
    uvw[torch.arange(n_obj), axis] = sign
    local = half[b] * uvw
    ln = torch.zeros(n_obj, 3)
    ln[torch.arange(n_obj), axis] = sign
    cy, sy = torch.cos(yaw[b]), torch.sin(yaw[b])
    rot = lambda v: torch.stack([cy * v[:, 0] - sy * v[:, 1], sy * v[:, 0] + cy * v[:, 1], v[:, 2]], dim=-1)  # noqa: E731
    obj = centers[b] + rot(local)
    obj_n = rot(ln) + 0.03 * torch.randn(n_obj, 3, generator=g)
    # floaters
    fl = torch.stack([(torch.rand(n_float, generator=g) * 2 - 1) * extent, (torch.rand(n_float, generator=g) * 2 - 1) * extent,
                      torch.rand(n_float, generator=g) * 0.8], dim=-1)
    means = torch.cat([ground, obj, fl], 0)
    lo, hi = math.log(scale_lo), math.log(scale_hi)
    scales = lo + (hi - lo) * torch.rand(n, 3, generator=g)
    scales[:, 2] += math.log(0.1)
    quats = torch.cat([_quats_from_normals(torch.cat([ground_n, obj_n], 0), g), random_quat_tensor(n_float, g)], 0)
    n_surf = n_ground + n_obj
    opaque = torch.rand(n_surf, generator=g) < opaque_frac
    opac_s = torch.where(opaque, 3.0 + torch.randn(n_surf, generator=g), -2.5 + torch.randn(n_surf, generator=g))
    opac = torch.cat([opac_s, -3.0 + torch.randn(n_float, generator=g)], 0)[:, None]
    dc, rest = _sh_params(n, sh_degree, g)
    perm = torch.randperm(n, generator=g)
    return dict(means=means[perm].contiguous(), scales=scales[perm].contiguous(), quats=quats[perm].contiguous(),
                features_dc=dc[perm].contiguous(), features_rest=rest[perm].contiguous(), opacities=opac[perm].contiguous())


def ring_cameras(n_views: int = 8, radius: float = 9.0, height: float = 3.5, target=(0.0, 0.0, 0.3), width: int = 1920,
                 height_px: int = 1080, focal: float = 1500.0) -> list:
    """Cameras on a horizontal ring OUTSIDE the content of ``surface_dense_scene``, looking at its centre."""
    cams = []
    for i in range(n_views):
        az = 2 * math.pi * i / n_views
        eye = torch.tensor([radius * math.cos(az), radius * math.sin(az), height])
        cams.append(Camera(look_at_c2w(eye, torch.tensor(target)), focal, focal, width / 2.0, height_px / 2.0, width,
                           height_px))
    return cams


def touch_patches_on_blob(n_patches: int = 5, pts_per_patch: int = 2000, radius: float = 0.1, seed: int = 0,
                          extent=(0.0101, 0.0076), depth: float = 0.01):
    """Config #3's tactile supervision: ``n_patches`` gel-sensor contact patches on the surface of the object blob,
    in the dictionary layout the reference's dataparser hands to the model
    (dn_splatter/data/normal_nerfstudio.py:593-683): ``points_xyz`` [n,3] (a regular grid on the tangent plane: the
    gel image is 320x240 pixels at 6.34e-5 m, every 5th point kept), ``normals`` [n,3] (contact normals: outward),
    ``bbox`` [8,3] — the oriented box around the patch, vertex index = 4*xi + 2*yi + zi, reaching ``depth`` behind
    the contact plane (z_diff*5 there)."""
    g = torch.Generator().manual_seed(seed + 991)
    patches = []
    nx = int(round(math.sqrt(pts_per_patch * extent[0] / extent[1])))
    ny = max(pts_per_patch // nx, 1)
    for _ in range(n_patches):
        d = torch.nn.functional.normalize(torch.randn(3, generator=g), dim=0)
        d[2] = d[2].abs() * 0.5 + 0.1  # upper half: visible from the camera ring
        d = torch.nn.functional.normalize(d, dim=0)
        centre = radius * d
        a = torch.linalg.cross(d, torch.tensor([0.0, 0.0, 1.0]))
        a = torch.nn.functional.normalize(a, dim=0)
        b = torch.linalg.cross(d, a)
        u = torch.linspace(-extent[0], extent[0], nx)
        v = torch.linspace(-extent[1], extent[1], ny)
        uu, vv = torch.meshgrid(u, v, indexing="ij")
        pts = centre + uu.reshape(-1, 1) * a + vv.reshape(-1, 1) * b
        normals = d.expand_as(pts).clone() + 0.05 * torch.randn(pts.shape, generator=g)
        normals = torch.nn.functional.normalize(normals, dim=-1)
        corners = []
        for i in range(8):
            xi, yi, zi = (i >> 2) & 1, (i >> 1) & 1, i & 1
            corners.append(centre + (2 * xi - 1) * extent[0] * a + (2 * yi - 1) * extent[1] * b + (1.05 * zi - 1) * depth * d)
        patches.append(dict(points_xyz=pts.contiguous(), points_rgb=torch.zeros_like(pts), normals=normals,
                            bbox=torch.stack(corners)))
    return patches
