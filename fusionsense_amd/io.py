"""On-disk formats either side of the hot path, host-side only.

IN   seed point clouds (`merged_pcd.ply` / `foreground_pcd.ply`, read with open3d at
     /root/reference/dn_splatter/data/normal_nerfstudio.py:97-135, 593-685) -> ``read_ply`` / ``read_pcd``
     and the Gaussian initialisation of ``DNSplatterModel.populate_modules`` (dn_model.py:196-300)
     -> ``init_gauss_params``.
OUT  the 3DGS PLY that ``ns-export gaussian-splat`` writes for a splatfacto model (scripts/train.py:215-220
     points at it; layout recalled from nerfstudio 1.1.3 ``ExportGaussianSplat``):
     float32 vertex properties x y z nx ny nz f_dc_0..2 f_rest_0..3(K-1)-1 opacity scale_0..2 rot_0..3,
     f_rest channel-major (``features_rest.transpose(1, 2)``), raw (un-activated) opacity / scale,
     rows with a non-finite value dropped -> ``export_gaussian_splat_ply`` (+ ``import_…`` for round trips);
     and nerfstudio's checkpoint dictionary (``step`` / ``pipeline`` with ``_model.gauss_params.<name>`` /
     ``optimizers``; ``SplatfactoModel.load_state_dict`` resizes the parameters to the stored shapes first)
     -> ``save_checkpoint`` / ``load_checkpoint``.
No open3d / plyfile here: the two point-cloud formats are parsed with numpy."""
from __future__ import annotations

import io as _io
from typing import Dict, Optional

import numpy as np
import torch
from torch import Tensor

from .scenes import random_quat_tensor, rgb_to_sh

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
              "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
              "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply(path: str) -> Dict[str, np.ndarray]:
    """Vertex properties of an ascii / binary_little_endian / binary_big_endian PLY as {name: [N] array}.
    Other elements (faces) are skipped when they follow the vertices; list properties are not supported."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n_vertex, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vertex = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(_io.BytesIO(b"".join(f.readline() for _ in range(n_vertex))), ndmin=2)
            if rows.shape[0] != n_vertex:
                raise ValueError(f"{path}: expected {n_vertex} vertices, found {rows.shape[0]}")
            return {name: rows[:, i].astype(np.dtype(t)) for i, (name, t) in enumerate(props)}
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(name, end + t) for name, t in props])
        data = np.frombuffer(f.read(n_vertex * dt.itemsize), dtype=dt, count=n_vertex)
        return {name: np.ascontiguousarray(data[name]).astype(np.dtype(t)) for name, t in props}


def write_ply(path: str, props: Dict[str, np.ndarray], comment: Optional[str] = None) -> None:
    """binary_little_endian PLY with one vertex element; property order = dict order."""
    names = list(props)
    n = len(props[names[0]]) if names else 0
    inv = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float",
           "f8": "double"}
    dt = np.dtype([(k, "<" + np.asarray(props[k]).dtype.str[1:]) for k in names])
    arr = np.empty(n, dtype=dt)
    for k in names:
        if len(props[k]) != n:
            raise ValueError("all properties need the same length")
        arr[k] = props[k]
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n")
        if comment:
            f.write(f"comment {comment}\n".encode())
        f.write(f"element vertex {n}\n".encode())
        for k in names:
            f.write(f"property {inv[np.asarray(props[k]).dtype.str[1:]]} {k}\n".encode())
        f.write(b"end_header\n")
        f.write(arr.tobytes())


def read_pcd(path: str) -> Dict[str, np.ndarray]:
    """PCD v0.7 (ascii or binary, uncompressed) as {field: [N] array} — the touch patches' format."""
    with open(path, "rb") as f:
        fields, sizes, types, counts, n, data = [], [], [], [], 0, None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PCD header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0].startswith("#"):
                continue
            key = tok[0].upper()
            if key == "FIELDS":
                fields = tok[1:]
            elif key == "SIZE":
                sizes = [int(t) for t in tok[1:]]
            elif key == "TYPE":
                types = tok[1:]
            elif key == "COUNT":
                counts = [int(t) for t in tok[1:]]
            elif key == "POINTS":
                n = int(tok[1])
            elif key == "DATA":
                data = tok[1].lower()
                break
        counts = counts or [1] * len(fields)
        if any(c != 1 for c in counts):
            raise ValueError(f"{path}: multi-count PCD fields are not supported")
        code = {"F": "f", "I": "i", "U": "u"}
        dts = [np.dtype(f"<{code[t.upper()]}{s}") for t, s in zip(types, sizes)]
        if data == "ascii":
            rows = np.loadtxt(f, ndmin=2)[:n]
            return {name: rows[:, i].astype(d) for i, (name, d) in enumerate(zip(fields, dts))}
        if data != "binary":
            raise ValueError(f"{path}: PCD DATA {data} is not supported")
        dt = np.dtype([(name, d) for name, d in zip(fields, dts)])
        arr = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
        return {name: np.ascontiguousarray(arr[name]) for name in fields}


def load_seed_points(path: str):
    """(points [N,3] f32, colours [N,3] f32 in 0..255, normals [N,3] f32 or None) of a seed PLY — the triple the
    dataparser hands to the model as ``seed_points`` (normal_nerfstudio.py:97-135)."""
    p = read_ply(path)
    pts = np.stack([p["x"], p["y"], p["z"]], -1).astype(np.float32)
    if all(k in p for k in ("red", "green", "blue")):
        col = np.stack([p["red"], p["green"], p["blue"]], -1).astype(np.float32)
        if np.issubdtype(p["red"].dtype, np.floating) and col.max(initial=0.0) <= 1.0:
            col = col * 255.0  # open3d normalises colours to [0,1]; the dataparser scales them back
    else:
        col = np.full_like(pts, 127.5)
    nrm = None
    if all(k in p for k in ("nx", "ny", "nz")):
        nrm = np.stack([p["nx"], p["ny"], p["nz"]], -1).astype(np.float32)
    return torch.from_numpy(pts), torch.from_numpy(col), (torch.from_numpy(nrm) if nrm is not None else None)


def knn_mean_distance(points: Tensor, k: int = 3) -> Tensor:
    """``k_nearest_sklearn(means, 3)`` then the mean over the k neighbours (dn_model.py:250-253): sklearn's
    NearestNeighbors(k+1, metric euclidean) with the self match dropped.  [N,1] float32."""
    from sklearn.neighbors import NearestNeighbors
    x = points.detach().cpu().numpy().astype(np.float32)
    nn = NearestNeighbors(n_neighbors=min(k + 1, len(x)), algorithm="auto", metric="euclidean").fit(x)
    dist, _ = nn.kneighbors(x)
    return torch.from_numpy(dist[:, 1:].astype(np.float32)).mean(dim=-1, keepdim=True)


def init_gauss_params(points: Tensor, colors_255: Tensor, normals: Optional[Tensor] = None, sh_degree: int = 3,
                      generator=None) -> Dict[str, Tensor]:
    """``populate_modules`` (dn_model.py:196-300) for seed points: means = points; SH dc = RGB2SH(colour/255),
    higher bands 0; opacity logit(0.1); scales = log(mean distance to the 3 nearest neighbours) on all axes —
    with seed normals the third axis is 10x thinner and the quaternion turns +z onto the normal; otherwise
    random quaternions."""
    from .touch import matrix_to_quaternion, rotate_vector_to_vector
    n = points.shape[0]
    K = (sh_degree + 1) ** 2
    avg = knn_mean_distance(points, 3)
    scales = torch.log(avg.repeat(1, 3))
    if normals is not None:
        nrm = normals.float()
        nrm = nrm / torch.norm(nrm, dim=-1, keepdim=True)
        scales[:, 2] = torch.log((avg / 10)[:, 0])
        mat = rotate_vector_to_vector(torch.tensor([0.0, 0.0, 1.0]).repeat(n, 1), nrm)
        quats = matrix_to_quaternion(mat)
    else:
        quats = random_quat_tensor(n, generator)
    if sh_degree > 0:
        dc = rgb_to_sh(colors_255.float() / 255)
    else:
        dc = torch.logit(colors_255.float() / 255, eps=1e-10)
    return {"means": points.float().clone(), "scales": scales, "quats": quats.float(), "features_dc": dc,
            "features_rest": torch.zeros(n, K - 1, 3), "opacities": torch.logit(0.1 * torch.ones(n, 1))}


def export_gaussian_splat_ply(gauss_params: Dict[str, Tensor], path: str) -> int:
    """Write the splat PLY of ``ns-export gaussian-splat``; returns the number of rows written (rows with
    NaN / Inf anywhere are dropped, as the exporter does)."""
    g = {k: v.detach().cpu().float() for k, v in gauss_params.items()}
    n = g["means"].shape[0]
    cols: Dict[str, np.ndarray] = {}
    xyz = g["means"].numpy()
    for i, k in enumerate("xyz"):
        cols[k] = xyz[:, i]
    for k in ("nx", "ny", "nz"):
        cols[k] = np.zeros(n, dtype=np.float32)
    dc = g["features_dc"].numpy()
    for i in range(3):
        cols[f"f_dc_{i}"] = dc[:, i]
    rest = g["features_rest"].transpose(1, 2).reshape(n, -1).numpy()  # channel-major, like the 3DGS viewers expect
    for i in range(rest.shape[1]):
        cols[f"f_rest_{i}"] = rest[:, i]
    cols["opacity"] = g["opacities"].reshape(n).numpy()
    sc = g["scales"].numpy()
    for i in range(3):
        cols[f"scale_{i}"] = sc[:, i]
    q = g["quats"].numpy()
    for i in range(4):
        cols[f"rot_{i}"] = q[:, i]
    keep = np.ones(n, dtype=bool)
    for v in cols.values():
        keep &= np.isfinite(v)
    cols = {k: np.ascontiguousarray(v[keep]).astype(np.float32) for k, v in cols.items()}
    write_ply(path, cols, comment="Generated by fusionsense_amd (ns-export gaussian-splat layout)")
    return int(keep.sum())


def import_gaussian_splat_ply(path: str) -> Dict[str, Tensor]:
    p = read_ply(path)
    n = len(p["x"])
    n_rest = sum(1 for k in p if k.startswith("f_rest_"))
    t = lambda names: torch.from_numpy(np.stack([p[k] for k in names], -1).astype(np.float32))  # noqa: E731
    rest = t([f"f_rest_{i}" for i in range(n_rest)]).reshape(n, 3, n_rest // 3).transpose(1, 2).contiguous() \
        if n_rest else torch.zeros(n, 0, 3)
    return {"means": t("xyz"), "scales": t([f"scale_{i}" for i in range(3)]), "quats": t([f"rot_{i}" for i in range(4)]),
            "features_dc": t([f"f_dc_{i}" for i in range(3)]), "features_rest": rest,
            "opacities": t(["opacity"])}


_CKPT_PREFIX = "_model.gauss_params."


def save_checkpoint(trainer, path: str) -> None:
    """nerfstudio's ``step-XXXXXXXXX.ckpt`` dictionary for the Gaussian model state: the parameters under
    ``pipeline`` as ``_model.gauss_params.<name>`` and one optimizer state dict per parameter group."""
    if hasattr(trainer, "flush"):
        trainer.flush()  # a deferred feature update (data-parallel runs) must be in the parameters
    if hasattr(trainer, "sync_optimizer_state"):
        trainer.sync_optimizer_state()  # (sharded geometry step: a collective — every rank saves or none)
    pipeline = {_CKPT_PREFIX + k: v.detach().cpu() for k, v in trainer.params.items()}
    # gauss_params["normals"] is re-derived by every get_outputs (dn_model.py:634) but is a key of the reference's
    # state dict: the last frame's world-space normals, or zeros before the first frame
    n = trainer.params["means"].shape[0]
    nw = getattr(getattr(trainer, "last_info", None), "normals_world", None)
    pipeline[_CKPT_PREFIX + "normals"] = (nw.detach().cpu() if nw is not None and nw.shape[0] == n
                                          else torch.zeros(n, 3))
    optimizers = {k: opt.state_dict() for k, opt in trainer.optimizers.items()}
    # nerfstudio stores the step it has just COMPLETED and resumes at step + 1; trainer.step counts completed steps
    torch.save({"step": max(int(trainer.step) - 1, 0), "fsgs_next_step": int(trainer.step), "pipeline": pipeline,
                "optimizers": optimizers, "adam_steps": int(getattr(trainer, "adam_steps", 0))}, path)


def load_checkpoint(trainer, path: str) -> int:
    """Inverse of ``save_checkpoint``; like ``SplatfactoModel.load_state_dict`` it first resizes every
    parameter (and with it the gradient slab and the optimizers) to the stored shapes.  Returns the step."""
    if hasattr(trainer, "flush"):
        trainer.flush()
    from .trainer import PARAM_ORDER, GradSlab
    ck = torch.load(path, map_location="cpu", weights_only=False)
    dev = trainer.device
    for name in PARAM_ORDER:
        v = ck["pipeline"][_CKPT_PREFIX + name].to(device=dev, dtype=torch.float32).contiguous()
        old = trainer.params[name]
        new = torch.nn.Parameter(v)
        trainer.params[name] = new
        opt = trainer.optimizers[name]
        opt.param_groups[0]["params"] = [new]
        opt.state.pop(old, None)
    trainer.slab.rebuild(trainer.params)
    if hasattr(trainer, "mark_params_written"):
        trainer.mark_params_written()
    for name in PARAM_ORDER:
        sd = ck["optimizers"].get(name)
        if sd is not None:
            trainer.optimizers[name].load_state_dict(sd)
            for st in trainer.optimizers[name].state.values():
                for k2, t2 in list(st.items()):
                    if torch.is_tensor(t2) and k2 != "step":
                        st[k2] = t2.to(dev)
    if getattr(trainer, "_geo", None) is not None:
        trainer._flatten_geometry()  # (load_state_dict replaced the moment tensors: back into the flat slabs)
    # a nerfstudio-written file has only "step" (the completed one): resume at step + 1, as its Trainer does
    trainer.step = int(ck["fsgs_next_step"]) if "fsgs_next_step" in ck else int(ck["step"]) + 1
    if "adam_steps" in ck:
        trainer.adam_steps = int(ck["adam_steps"])
    else:  # the bias-correction count of warm moments comes from the optimizer state, not from zero
        steps = [float(st["step"]) for opt in trainer.optimizers.values() for st in opt.state.values() if "step" in st]
        trainer.adam_steps = int(max(steps)) if steps else 0
    strat = getattr(trainer, "strategy", None)
    if strat is not None:  # statistics are not part of a nerfstudio checkpoint: they restart
        strat.xys_grad_norm = strat.vis_counts = strat.max_2Dsize = None
    return trainer.step
