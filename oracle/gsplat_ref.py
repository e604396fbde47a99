"""CPU oracle for the Gaussian-splatting hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain PyTorch/numpy restatement of the arithmetic that FusionSense
obtains from the un-vendored packages ``gsplat==1.0.0`` (reference pin:
pyproject.toml:8, env1.yml:283) at its call sites dn_splatter/dn_model.py:570-591
(``rasterization``) and dn_splatter/dn_model.py:644-653 (legacy ``rasterize_gaussians``).

PARITY UNPINNED: the reference repository vendors neither gsplat nor nerfstudio and ships
no tests or golden vectors for this path, so this restatement follows
the published gsplat 1.0.0 algorithm as recorded in SURVEY.md §8a / Appendix A, anchored on
the reference's own call sites.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never does.

All functions take a ``dtype`` implicitly from their inputs (fp32 or fp64) and run on CPU.
Integer outputs (radii, tile counts, intersection keys, sorted ids, offsets, last ids)
are produced with numpy integer arithmetic so that they can be compared bit-for-bit.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

ALPHA_MAX = 0.999  # gsplat rasterize_to_pixels: alpha = min(0.999, opac * exp(-sigma))
ALPHA_MIN = 1.0 / 255.0  # skip if alpha < 1/255
T_MIN = 1e-4  # stop before the Gaussian that would bring T to <= 1e-4


# --------------------------------------------------------------------------------------
# E9: helpers on the import surface (dn_model.py:34-35)
# --------------------------------------------------------------------------------------
def num_sh_bases(degree: int) -> int:
    """gsplat.cuda_legacy._wrapper.num_sh_bases (used dn_model.py:205,286)."""
    if degree < 0 or degree > 4:
        raise AssertionError("gsplat supports SH degree 0..4")
    return (degree + 1) ** 2


def quat_to_rotmat(quat: Tensor) -> Tensor:
    """gsplat.cuda_legacy._torch_impl.quat_to_rotmat: wxyz, input is normalised first
    (SURVEY.md A.4).  Used at dn_model.py:623,1191,1489,1699,1770,2146."""
    assert quat.shape[-1] == 4, quat.shape
    w, x, y, z = torch.unbind(torch.nn.functional.normalize(quat, dim=-1), dim=-1)
    mat = torch.stack(
        [
            1 - 2 * (y**2 + z**2),
            2 * (x * y - w * z),
            2 * (x * z + w * y),
            2 * (x * y + w * z),
            1 - 2 * (x**2 + z**2),
            2 * (y * z - w * x),
            2 * (x * z - w * y),
            2 * (y * z + w * x),
            1 - 2 * (x**2 + y**2),
        ],
        dim=-1,
    )
    return mat.reshape(quat.shape[:-1] + (3, 3))


def get_viewmat(c2w: Tensor) -> Tensor:
    """nerfstudio.models.splatfacto.get_viewmat (used dn_model.py:550): OpenGL c2w [B,3,4]
    -> OpenCV world-to-camera [B,4,4]."""
    R = c2w[:, :3, :3]
    T = c2w[:, :3, 3:4]
    R = R * torch.tensor([[[1.0, -1.0, -1.0]]], dtype=R.dtype)
    R_inv = R.transpose(1, 2)
    T_inv = -torch.bmm(R_inv, T)
    viewmat = torch.zeros(R.shape[0], 4, 4, dtype=R.dtype)
    viewmat[:, 3, 3] = 1.0
    viewmat[:, :3, :3] = R_inv
    viewmat[:, :3, 3:4] = T_inv
    return viewmat


# --------------------------------------------------------------------------------------
# E2: fully_fused_projection
# --------------------------------------------------------------------------------------
def quat_scale_to_covar(quats: Tensor, scales: Tensor) -> Tensor:
    R = quat_to_rotmat(quats)
    M = R * scales[..., None, :]
    return M @ M.transpose(-1, -2)


def project(
    means: Tensor,  # [N,3]
    quats: Tensor,  # [N,4]
    scales: Tensor,  # [N,3]
    viewmats: Tensor,  # [C,4,4]
    Ks: Tensor,  # [C,3,3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Returns radii[C,N] (int32), means2d[C,N,2], depths[C,N], conics[C,N,3],
    compensations[C,N].  Outputs of culled Gaussians are zero (radii == 0 marks them)."""
    dt = means.dtype
    covars = quat_scale_to_covar(quats, scales)  # [N,3,3]
    R = viewmats[:, :3, :3]  # [C,3,3]
    t = viewmats[:, :3, 3]  # [C,3]
    means_c = torch.einsum("cij,nj->cni", R, means) + t[:, None, :]  # [C,N,3]
    covars_c = torch.einsum("cij,njk,clk->cnil", R, covars, R)  # [C,N,3,3]

    tx, ty, tz = torch.unbind(means_c, dim=-1)
    fx = Ks[:, 0, 0][:, None]
    fy = Ks[:, 1, 1][:, None]
    cx = Ks[:, 0, 2][:, None]
    cy = Ks[:, 1, 2][:, None]
    lim_x = 1.3 * (0.5 * width / fx)
    lim_y = 1.3 * (0.5 * height / fy)
    tz_safe = torch.where(tz == 0, torch.ones_like(tz), tz)
    txc = tz * torch.minimum(torch.maximum(tx / tz_safe, -lim_x), lim_x)
    tyc = tz * torch.minimum(torch.maximum(ty / tz_safe, -lim_y), lim_y)
    O = torch.zeros_like(tz)
    J = torch.stack(
        [fx / tz_safe, O, -fx * txc / tz_safe**2, O, fy / tz_safe, -fy * tyc / tz_safe**2],
        dim=-1,
    ).reshape(tz.shape + (2, 3))
    cov2d = J @ covars_c @ J.transpose(-1, -2)  # [C,N,2,2]
    means2d = torch.stack([fx * tx / tz_safe + cx, fy * ty / tz_safe + cy], dim=-1)

    a0 = cov2d[..., 0, 0]
    b0 = 0.5 * (cov2d[..., 0, 1] + cov2d[..., 1, 0])
    c0 = cov2d[..., 1, 1]
    det_orig = a0 * c0 - b0 * b0
    a = a0 + eps2d
    c = c0 + eps2d
    det = a * c - b0 * b0
    det_safe = torch.where(det > 0, det, torch.ones_like(det))
    comp = torch.sqrt(torch.clamp(det_orig / det_safe, min=0.0))
    conics = torch.stack([c / det_safe, -b0 / det_safe, a / det_safe], dim=-1)
    bb = 0.5 * (a + c)
    v1 = bb + torch.sqrt(torch.clamp(bb * bb - det, min=0.01))
    radius = torch.ceil(3.0 * torch.sqrt(v1))

    valid = (tz >= near_plane) & (tz <= far_plane) & (det > 0) & (radius > radius_clip)
    mx, my = means2d[..., 0], means2d[..., 1]
    inside = (mx + radius > 0) & (mx - radius < width) & (my + radius > 0) & (my - radius < height)
    valid = valid & inside
    radii = torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32)
    vf = valid.to(dt)
    return (
        radii,
        means2d * vf[..., None],
        tz * vf,
        conics * vf[..., None],
        comp * vf,
    )


# --------------------------------------------------------------------------------------
# E3: spherical harmonics
# --------------------------------------------------------------------------------------
def sh_bases(degree: int, dirs: Tensor) -> Tensor:
    """Real SH basis values [..., (degree+1)^2] for *normalised* directions (Sloan's
    efficient evaluation, the form gsplat's kernel uses)."""
    x, y, z = torch.unbind(dirs, dim=-1)
    out = [torch.full_like(x, 0.2820947917738781)]
    if degree >= 1:
        out += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
    if degree >= 2:
        z2 = z * z
        fTmp0B = -1.092548430592079 * z
        fC1 = x * x - y * y
        fS1 = 2.0 * x * y
        out += [
            0.5462742152960395 * fS1,
            fTmp0B * y,
            0.9461746957575601 * z2 - 0.3153915652525201,
            fTmp0B * x,
            0.5462742152960395 * fC1,
        ]
    if degree >= 3:
        fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658
        fTmp1B = 1.445305721320277 * z
        fC2 = x * fC1 - y * fS1
        fS2 = x * fS1 + y * fC1
        pSH12 = z * (1.865881662950577 * z2 - 1.119528997770346)
        out += [
            -0.5900435899266435 * fS2,
            fTmp1B * fS1,
            fTmp0C * y,
            pSH12,
            fTmp0C * x,
            fTmp1B * fC1,
            -0.5900435899266435 * fC2,
        ]
    if degree >= 4:
        fTmp0D = z * (-4.683325804901025 * z2 + 2.007139630671868)
        fTmp1C = 3.31161143515146 * z2 - 0.47308734787878
        fTmp2B = -1.770130769779931 * z
        fC3 = x * fC2 - y * fS2
        fS3 = x * fS2 + y * fC2
        pSH6 = 0.9461746957575601 * z2 - 0.3153915652525201
        out += [
            0.6258357354491763 * fS3,
            fTmp2B * fS2,
            fTmp1C * fS1,
            fTmp0D * y,
            1.984313483298443 * z * pSH12 - 1.006230589874905 * pSH6,
            fTmp0D * x,
            fTmp1C * fC1,
            fTmp2B * fC2,
            0.6258357354491763 * fC3,
        ]
    return torch.stack(out, dim=-1)


def spherical_harmonics(
    degree: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor] = None
) -> Tensor:
    """dirs [...,3] (unnormalised), coeffs [...,K,3] -> colours [...,3]; zero where masked
    out.  Only the first (degree+1)^2 bases are used."""
    d = torch.nn.functional.normalize(dirs, dim=-1)
    B = sh_bases(degree, d)  # [...,k]
    k = B.shape[-1]
    col = (B[..., None] * coeffs[..., :k, :]).sum(dim=-2)
    if masks is not None:
        col = col * masks[..., None].to(col.dtype)
    return col


# --------------------------------------------------------------------------------------
# E4-E6: tile intersection, sort, offsets.  Integer work -> numpy.
# --------------------------------------------------------------------------------------
def tile_bits(n_tiles: int) -> int:
    return int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1


def tile_bbox(
    means2d: np.ndarray, radii: np.ndarray, tile_size: int, tw: int, th: int, legacy: bool = False
):
    """Per-Gaussian tile rectangle [min, max) in tile units.  New path: floor/ceil;
    legacy (cuda_legacy get_tile_bbox): (int)(c-r), (int)(c+r+1).  Clamp BEFORE casting
    (SURVEY.md A.4)."""
    m = means2d.astype(np.float32) / np.float32(tile_size)
    r = radii.astype(np.float32) / np.float32(tile_size)
    lo = m - r[..., None]
    hi = m + r[..., None]
    if legacy:
        lo_i = np.trunc(lo)
        hi_i = np.trunc(hi + np.float32(1.0))
    else:
        lo_i = np.floor(lo)
        hi_i = np.ceil(hi)
    lim = np.array([tw, th], dtype=np.float32)
    lo_i = np.clip(lo_i, 0, lim).astype(np.int64)
    hi_i = np.clip(hi_i, 0, lim).astype(np.int64)
    return lo_i, hi_i


def isect_tiles(
    means2d: Tensor,  # [C,N,2] fp32
    radii: Tensor,  # [C,N] int32
    depths: Tensor,  # [C,N] fp32
    tile_size: int,
    tw: int,
    th: int,
    legacy: bool = False,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Returns tiles_per_gauss [C,N] int32, isect_ids [M] int64 (UNSORTED, emission order),
    flatten_ids [M] int32."""
    m2 = means2d.detach().to(torch.float32).numpy()
    rd = radii.detach().numpy().astype(np.int32)
    dp = depths.detach().to(torch.float32).numpy()
    C, N = rd.shape
    lo, hi = tile_bbox(m2, rd, tile_size, tw, th, legacy)
    cnt = (hi[..., 0] - lo[..., 0]) * (hi[..., 1] - lo[..., 1])
    cnt = np.where(rd > 0, cnt, 0).astype(np.int32)
    M = int(cnt.sum())
    n_tiles = tw * th
    tb = tile_bits(n_tiles)
    depth_bits = dp.view(np.int32).astype(np.int64) & 0xFFFFFFFF
    ids = np.empty(M, dtype=np.int64)
    flat = np.empty(M, dtype=np.int32)
    cntf = cnt.reshape(-1)
    cum = np.concatenate([[0], np.cumsum(cntf.astype(np.int64))])
    lo_f = lo.reshape(-1, 2)
    hi_f = hi.reshape(-1, 2)
    db_f = depth_bits.reshape(-1)
    for idx in np.nonzero(cntf)[0]:
        c = idx // N
        ys = np.arange(lo_f[idx, 1], hi_f[idx, 1], dtype=np.int64)
        xs = np.arange(lo_f[idx, 0], hi_f[idx, 0], dtype=np.int64)
        tid = (ys[:, None] * tw + xs[None, :]).reshape(-1)
        s = cum[idx]
        ids[s : s + tid.size] = (np.int64(c) << (32 + tb)) | (tid << 32) | db_f[idx]
        flat[s : s + tid.size] = idx
    return cnt, ids, flat


def sort_isects(isect_ids: np.ndarray, flatten_ids: np.ndarray):
    """Stable sort by key == LSD radix order (ties keep ascending emission order)."""
    order = np.argsort(isect_ids.view(np.uint64), kind="stable")
    return isect_ids[order], flatten_ids[order]


def isect_offset_encode(isect_ids_sorted: np.ndarray, C: int, tw: int, th: int) -> np.ndarray:
    n_tiles = tw * th
    tb = tile_bits(n_tiles)
    hi = isect_ids_sorted >> 32
    cid = hi >> tb
    tid = hi & ((1 << tb) - 1)
    lin = cid * n_tiles + tid
    q = np.arange(C * n_tiles, dtype=np.int64)
    return np.searchsorted(lin, q, side="left").astype(np.int32).reshape(C, th, tw)


# --------------------------------------------------------------------------------------
# E7: rasterize_to_pixels forward / backward
# --------------------------------------------------------------------------------------
def _tile_pixels(ty: int, tx: int, tile_size: int, W: int, H: int, dtype):
    y0, x0 = ty * tile_size, tx * tile_size
    y1, x1 = min(y0 + tile_size, H), min(x0 + tile_size, W)
    ys = torch.arange(y0, y1)
    xs = torch.arange(x0, x1)
    py = (ys.to(dtype) + 0.5)[:, None].expand(len(ys), len(xs)).reshape(-1)
    px = (xs.to(dtype) + 0.5)[None, :].expand(len(ys), len(xs)).reshape(-1)
    return (y0, y1, x0, x1), px, py


def _tile_forward(px, py, xy, conic, opac):
    """Per-tile dense evaluation.  Returns (a_eff, vis, T_excl, alive_valid) each [P,n]."""
    dx = xy[None, :, 0] - px[:, None]
    dy = xy[None, :, 1] - py[:, None]
    sigma = 0.5 * (conic[None, :, 0] * dx * dx + conic[None, :, 2] * dy * dy) + conic[None, :, 1] * dx * dy
    vis = torch.exp(-sigma)
    alpha = torch.clamp(opac[None, :] * vis, max=ALPHA_MAX)
    valid = (sigma >= 0) & (alpha >= ALPHA_MIN)
    a = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1.0 - a
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    alive = T_incl > T_MIN  # monotone: first failure terminates the pixel
    contrib = valid & alive
    a_eff = torch.where(contrib, a, torch.zeros_like(a))
    return a_eff, vis, T_excl, contrib, dx, dy


def rasterize_to_pixels(
    means2d: Tensor,  # [C,N,2]
    conics: Tensor,  # [C,N,3]
    colors: Tensor,  # [C,N,D]
    opacities: Tensor,  # [C,N]
    width: int,
    height: int,
    tile_size: int,
    isect_offsets: np.ndarray,  # [C,th,tw] int32
    flatten_ids: np.ndarray,  # [M] int32
    backgrounds: Optional[Tensor] = None,  # [C,D]
) -> Tuple[Tensor, Tensor, np.ndarray]:
    """Differentiable (torch autograd) front-to-back composite.  Returns render[C,H,W,D],
    alphas[C,H,W,1], last_ids[C,H,W] int32 (absolute index into flatten_ids)."""
    C, N = opacities.shape
    D = colors.shape[-1]
    dt = means2d.dtype
    th, tw = isect_offsets.shape[1:]
    M = flatten_ids.shape[0]
    offs = np.concatenate([isect_offsets.reshape(-1).astype(np.int64), [M]])
    m2 = means2d.reshape(C * N, 2)
    cn = conics.reshape(C * N, 3)
    cl = colors.reshape(C * N, D)
    op = opacities.reshape(C * N)
    fl = torch.from_numpy(flatten_ids.astype(np.int64))
    rows_rgb = []
    rows_alpha = []
    last_ids = np.zeros((C, height, width), dtype=np.int32)
    for c in range(C):
        strips_rgb, strips_a = [], []
        for ty in range(th):
            row_rgb, row_a = [], []
            for tx in range(tw):
                t_lin = (c * th + ty) * tw + tx
                s, e = int(offs[t_lin]), int(offs[t_lin + 1])
                (y0, y1, x0, x1), px, py = _tile_pixels(ty, tx, tile_size, width, height, dt)
                P = px.shape[0]
                if e > s:
                    g = fl[s:e]
                    a_eff, vis, T_excl, contrib, _, _ = _tile_forward(px, py, m2[g], cn[g], op[g])
                    w = a_eff * T_excl
                    pix = w @ cl[g]
                    T_fin = torch.prod(1.0 - a_eff, dim=1)
                    idx = torch.arange(s, e)[None, :].expand(P, -1)
                    li = torch.where(contrib, idx, torch.zeros_like(idx)).max(dim=1).values
                else:
                    # keep the graph connected so empty scenes still back-propagate zeros
                    pix = torch.zeros(P, D, dtype=dt) + 0.0 * cl.sum()
                    T_fin = torch.ones(P, dtype=dt)
                    li = torch.zeros(P, dtype=torch.int64)
                if backgrounds is not None:
                    pix = pix + T_fin[:, None] * backgrounds[c][None, :]
                row_rgb.append(pix.reshape(y1 - y0, x1 - x0, D))
                row_a.append((1.0 - T_fin).reshape(y1 - y0, x1 - x0, 1))
                last_ids[c, y0:y1, x0:x1] = li.reshape(y1 - y0, x1 - x0).numpy().astype(np.int32)
            strips_rgb.append(torch.cat(row_rgb, dim=1))
            strips_a.append(torch.cat(row_a, dim=1))
        rows_rgb.append(torch.cat(strips_rgb, dim=0))
        rows_alpha.append(torch.cat(strips_a, dim=0))
    return torch.stack(rows_rgb), torch.stack(rows_alpha), last_ids


@torch.no_grad()
def rasterize_to_pixels_bwd(
    means2d: Tensor,
    conics: Tensor,
    colors: Tensor,
    opacities: Tensor,
    width: int,
    height: int,
    tile_size: int,
    isect_offsets: np.ndarray,
    flatten_ids: np.ndarray,
    v_render: Tensor,  # [C,H,W,D]
    v_alpha: Tensor,  # [C,H,W,1]
    backgrounds: Optional[Tensor] = None,
) -> Dict[str, Tensor]:
    """Hand-derived VJP in the kernel's own form, needed because
    ``absgrad`` (sum over pixels of |v_xy|) is not an autograd quantity.  Checked against
    autograd of :func:`rasterize_to_pixels` in tests/test_oracle.py."""
    C, N = opacities.shape
    D = colors.shape[-1]
    dt = means2d.dtype
    th, tw = isect_offsets.shape[1:]
    M = flatten_ids.shape[0]
    offs = np.concatenate([isect_offsets.reshape(-1).astype(np.int64), [M]])
    m2 = means2d.reshape(C * N, 2)
    cn = conics.reshape(C * N, 3)
    cl = colors.reshape(C * N, D)
    op = opacities.reshape(C * N)
    fl = torch.from_numpy(flatten_ids.astype(np.int64))
    v_m2 = torch.zeros_like(m2)
    v_abs = torch.zeros_like(m2)
    v_cn = torch.zeros_like(cn)
    v_cl = torch.zeros_like(cl)
    v_op = torch.zeros_like(op)
    for c in range(C):
        for ty in range(th):
            for tx in range(tw):
                t_lin = (c * th + ty) * tw + tx
                s, e = int(offs[t_lin]), int(offs[t_lin + 1])
                if e <= s:
                    continue
                (y0, y1, x0, x1), px, py = _tile_pixels(ty, tx, tile_size, width, height, dt)
                g = fl[s:e]
                xy, conic, opac, col = m2[g], cn[g], op[g], cl[g]
                a_eff, vis, T_excl, contrib, dx, dy = _tile_forward(px, py, xy, conic, opac)
                vo = v_render[c, y0:y1, x0:x1].reshape(-1, D)
                va = v_alpha[c, y0:y1, x0:x1].reshape(-1)
                T_fin = torch.prod(1.0 - a_eff, dim=1)
                fac = a_eff * T_excl
                sdot = vo @ col.T  # [P,n] = sum_k col_k * v_out_k
                wv = fac * sdot
                suffix = torch.flip(torch.cumsum(torch.flip(wv, [1]), 1), [1]) - wv  # sum_{j>i}
                ra = 1.0 / (1.0 - a_eff)
                v_a = sdot * T_excl - suffix * ra + (T_fin * va)[:, None] * ra
                if backgrounds is not None:
                    v_a = v_a - (T_fin * (vo @ backgrounds[c]))[:, None] * ra
                cf = contrib.to(dt)
                unclamped = (opac[None, :] * vis <= ALPHA_MAX).to(dt) * cf
                v_sigma = -opac[None, :] * vis * v_a * unclamped
                gx = v_sigma * (conic[None, :, 0] * dx + conic[None, :, 1] * dy)
                gy = v_sigma * (conic[None, :, 1] * dx + conic[None, :, 2] * dy)
                v_m2.index_add_(0, g, torch.stack([gx.sum(0), gy.sum(0)], -1))
                v_abs.index_add_(0, g, torch.stack([gx.abs().sum(0), gy.abs().sum(0)], -1))
                v_cn.index_add_(
                    0,
                    g,
                    torch.stack(
                        [
                            (0.5 * v_sigma * dx * dx).sum(0),
                            (v_sigma * dx * dy).sum(0),
                            (0.5 * v_sigma * dy * dy).sum(0),
                        ],
                        -1,
                    ),
                )
                v_op.index_add_(0, g, (vis * v_a * unclamped).sum(0))
                v_cl.index_add_(0, g, (fac * cf).T @ vo)
    return {
        "v_means2d": v_m2.reshape(C, N, 2),
        "v_means2d_abs": v_abs.reshape(C, N, 2),
        "v_conics": v_cn.reshape(C, N, 3),
        "v_colors": v_cl.reshape(C, N, D),
        "v_opacities": v_op.reshape(C, N),
    }


# --------------------------------------------------------------------------------------
# E1: rasterization orchestration (call site dn_model.py:570-591)
# --------------------------------------------------------------------------------------
def rasterization(
    means: Tensor,
    quats: Tensor,
    scales: Tensor,
    opacities: Tensor,
    colors: Tensor,
    viewmats: Tensor,
    Ks: Tensor,
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
) -> Tuple[Tensor, Tensor, Dict]:
    assert render_mode in ("RGB", "D", "ED", "RGB+D", "RGB+ED"), render_mode
    N = means.shape[0]
    C = viewmats.shape[0]
    radii, means2d, depths, conics, comp = project(
        means, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip
    )
    opac = opacities[None, :].expand(C, N)
    if rasterize_mode == "antialiased":
        opac = opac * comp
    tw = math.ceil(width / tile_size)
    th = math.ceil(height / tile_size)
    tpg, ids, flat = isect_tiles(means2d, radii, depths, tile_size, tw, th)
    ids_s, flat_s = sort_isects(ids, flat)
    offsets = isect_offset_encode(ids_s, C, tw, th)

    if sh_degree is None:
        cols = colors[None].expand(C, -1, -1) if colors.dim() == 2 else colors
    else:
        campos = torch.inverse(viewmats)[:, :3, 3]
        dirs = means[None, :, :] - campos[:, None, :]
        shs = colors[None].expand(C, -1, -1, -1)
        cols = spherical_harmonics(sh_degree, dirs, shs, masks=radii > 0)
        cols = torch.clamp_min(cols + 0.5, 0.0)
    if render_mode in ("RGB+D", "RGB+ED"):
        cols = torch.cat([cols, depths[..., None]], dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, dtype=backgrounds.dtype)], dim=-1)
    elif render_mode in ("D", "ED"):
        cols = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, dtype=backgrounds.dtype)

    render, alphas, last_ids = rasterize_to_pixels(
        means2d, conics, cols, opac, width, height, tile_size, offsets, flat_s, backgrounds
    )
    if render_mode in ("ED", "RGB+ED"):
        render = torch.cat(
            [render[..., :-1], render[..., -1:] / alphas.clamp(min=1e-10)], dim=-1
        )
    meta = {
        "camera_ids": None,
        "gaussian_ids": None,
        "radii": radii,
        "means2d": means2d,
        "depths": depths,
        "conics": conics,
        "opacities": opac,
        "tile_width": tw,
        "tile_height": th,
        "tiles_per_gauss": torch.from_numpy(tpg),
        "isect_ids": torch.from_numpy(ids_s),
        "flatten_ids": torch.from_numpy(flat_s),
        "isect_offsets": torch.from_numpy(offsets),
        "last_ids": torch.from_numpy(last_ids),
        "colors": cols,
        "width": width,
        "height": height,
        "tile_size": tile_size,
        "n_cameras": C,
    }
    return render, alphas, meta


# --------------------------------------------------------------------------------------
# E8: legacy rasterize_gaussians (SURVEY.md §8a-8'; call site dn_model.py:644-653)
# --------------------------------------------------------------------------------------
def rasterize_gaussians(
    xys: Tensor,  # [N,2]
    depths: Tensor,  # [N]
    radii: Tensor,  # [N] int32
    conics: Tensor,  # [N,3]
    num_tiles_hit: Tensor,  # [N] int32 (unused for binning: see DESIGN.md, legacy overflow)
    colors: Tensor,  # [N,ch]
    opacity: Tensor,  # [N,1]
    img_height: int,
    img_width: int,
    block_width: int,
    background: Optional[Tensor] = None,
    return_alpha: bool = False,
):
    assert 1 < block_width <= 16, "block_width must be between 2 and 16"
    ch = colors.shape[-1]
    if background is None:
        background = torch.ones(ch, dtype=colors.dtype)
    assert background.shape[0] == ch
    tw = (img_width + block_width - 1) // block_width
    th = (img_height + block_width - 1) // block_width
    _, ids, flat = isect_tiles(xys[None], radii[None], depths[None], block_width, tw, th, legacy=True)
    ids_s, flat_s = sort_isects(ids, flat)
    offsets = isect_offset_encode(ids_s, 1, tw, th)
    out, alpha, _ = rasterize_to_pixels(
        xys[None], conics[None], colors[None], opacity.reshape(1, -1), img_width, img_height,
        block_width, offsets, flat_s, background[None],
    )
    if return_alpha:
        return out[0], alpha[0, ..., 0]
    return out[0]


# --------------------------------------------------------------------------------------
# a-11: per-Gaussian normals exactly as dn_model.py:618-636 computes them
# --------------------------------------------------------------------------------------
def gaussian_normals(quats: Tensor, log_scales: Tensor, means: Tensor, c2w: Tensor):
    """Returns (normals_world [N,3], normals_cam [N,3]); c2w is [3,4] (OpenGL)."""
    q = quats / quats.norm(dim=-1, keepdim=True)
    onehot = torch.nn.functional.one_hot(torch.argmin(log_scales, dim=-1), num_classes=3).to(q.dtype)
    rots = quat_to_rotmat(q)
    n = torch.bmm(rots, onehot[:, :, None]).squeeze(-1)
    n = torch.nn.functional.normalize(n, dim=1)
    viewdirs = -means.detach() + c2w.detach()[:3, 3][None, :]
    viewdirs = viewdirs / viewdirs.norm(dim=-1, keepdim=True)
    dots = (n * viewdirs).sum(-1)
    n = torch.where((dots < 0)[:, None], -n, n)
    return n, n @ c2w[:3, :3]
