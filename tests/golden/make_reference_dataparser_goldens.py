#!/usr/bin/env python3
"""Golden vectors made by EXECUTING the reference's own dataparser and dataset in this container:

  NormalNerfstudio._generate_dataparser_outputs   /root/reference/dn_splatter/data/normal_nerfstudio.py:136-725
        frames sorted by file name, then natural order, then (no split spec) the same permutation once more; the
        y / z camera-axis flip (:317); auto_scale_poses (:325-330); seed / hull points moved by transform + scale;
        touch patches (:593-685: centring, gel pitch, the patch's and the dataparser's transforms, .npy / .pcd masks,
        2- and 3-channel normals, the oriented box); applied_transform / applied_scale
  GDataset.__init__ / get_metadata / get_normal_image_from_path   dn_splatter/data/dn_dataset.py:21-243
        depth scale = depth_unit_scale_factor x dataparser_scale, normal PNG -> [0,1], opengl -> opencv flip

over a scene directory written by tests/helpers.py:write_scene_dir (two variants; the tests write the same directory
from the same seed, so only the reference's OUTPUTS travel).  What the two files import from packages that are not
installed is supplied by stand-ins, restated from nerfstudio 1.1.3 as recalled: the ``Nerfstudio``
base dataparser (``_get_fname``, ``_load_3D_points``), ``camera_utils.auto_orient_and_center_poses`` (methods "none"),
``Cameras`` (a record), ``get_depth_image_from_path`` (PIL instead of cv2, no resize needed), natsort, and open3d's
point-cloud READERS (a minimal PLY / PCD parser written here, independent of fusionsense_amd/io.py).  open3d's PCA
normal estimation cannot be executed: the stand-in returns a fixed function of the points, and the golden pins what
the reference DOES with the normals (the [3,4] transform, translation included, :107-113).

Runs only where /root/reference exists; writes tests/golden/reference_dataparser.npz (numeric arrays only).
"""
import dataclasses
import json
import os
import re
import sys
import tempfile
import types
from pathlib import Path

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import make_reference_goldens as base  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "reference_dataparser.npz")


def fake_pca_normals(points: np.ndarray) -> np.ndarray:
    """What the open3d stand-in's estimate_normals + normalize_normals leaves in pcd.normals (the tests patch
    fusionsense_amd.dataparser.estimate_normals with the same function)."""
    n = np.stack([np.sin(3.0 * points[:, 0]) + 0.3, np.cos(2.0 * points[:, 1]), points[:, 2] + 0.5], -1)
    return n / np.linalg.norm(n, axis=-1, keepdims=True)


# ---- independent minimal readers (the open3d stand-in) ------------------------------------------------------------
def _read_ply(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt, props, n = None, [], 0
        while True:
            ln = f.readline().decode().strip()
            if ln.startswith("format"):
                fmt = ln.split()[1]
            elif ln.startswith("element vertex"):
                n = int(ln.split()[2])
            elif ln.startswith("property"):
                _, ty, name = ln.split()
                props.append((name, {"float": "<f4", "float32": "<f4", "double": "<f8", "uchar": "u1", "uint8": "u1",
                                     "int": "<i4", "short": "<i2"}[ty]))
            elif ln == "end_header":
                break
        if fmt == "ascii":
            rows = np.loadtxt(f, ndmin=2)[:n]
            return {name: rows[:, i] for i, (name, _) in enumerate(props)}
        data = np.frombuffer(f.read(), dtype=np.dtype(props), count=n)
        return {name: data[name] for name, _ in props}


def _read_pcd(path):
    with open(path, "rb") as f:
        fields = None
        while True:
            ln = f.readline().decode().strip()
            if ln.startswith("FIELDS"):
                fields = ln.split()[1:]
            if ln.startswith("DATA"):
                assert ln.split()[1] == "ascii"
                break
        rows = np.loadtxt(f, ndmin=2)
    return {k: rows[:, i] for i, k in enumerate(fields)}


class _PointCloud:
    def __init__(self):
        self.points = np.zeros((0, 3))
        self.colors = np.zeros((0, 3))
        self.normals = np.zeros((0, 3))

    def estimate_normals(self, search_param=None):
        self.normals = fake_pca_normals(np.asarray(self.points, dtype=np.float64))

    def normalize_normals(self):
        self.normals = self.normals / np.linalg.norm(self.normals, axis=-1, keepdims=True)


def _o3d_standin():
    o3d = types.ModuleType("open3d")

    def read_point_cloud(path):
        d = _read_ply(path) if str(path).endswith(".ply") else _read_pcd(path)
        pc = _PointCloud()
        pc.points = np.stack([d["x"], d["y"], d["z"]], -1).astype(np.float64)
        if "red" in d:  # open3d holds colours as doubles in [0, 1]
            pc.colors = np.stack([d["red"], d["green"], d["blue"]], -1).astype(np.float64) / 255.0
        else:
            pc.colors = np.zeros((0, 3))
        return pc

    o3d.io = types.SimpleNamespace(read_point_cloud=read_point_cloud, write_point_cloud=lambda *a, **k: None)
    o3d.geometry = types.SimpleNamespace(PointCloud=_PointCloud,
                                         KDTreeSearchParamHybrid=lambda radius, max_nn: (radius, max_nn))
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a))
    return o3d


def _natsorted(seq):
    return sorted(seq, key=lambda s: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", str(s))])


@dataclasses.dataclass
class _Cameras:
    """nerfstudio Cameras as a record of what the dataparser hands over."""
    fx: object = None
    fy: object = None
    cx: object = None
    cy: object = None
    distortion_params: object = None
    height: object = None
    width: object = None
    camera_to_worlds: object = None
    camera_type: object = None
    metadata: object = None

    def __post_init__(self):
        n = self.camera_to_worlds.shape[0]

        def per_cam(v, dt):
            return v.to(dt).reshape(n, 1) if torch.is_tensor(v) else torch.full((n, 1), v, dtype=dt)
        self.fx, self.fy = per_cam(self.fx, torch.float32), per_cam(self.fy, torch.float32)
        self.cx, self.cy = per_cam(self.cx, torch.float32), per_cam(self.cy, torch.float32)
        self.height, self.width = per_cam(self.height, torch.int64), per_cam(self.width, torch.int64)

    def rescale_output_resolution(self, scaling_factor):
        assert scaling_factor == 1.0


def _install():
    base._install_stubs()
    sys.modules["open3d"] = _o3d_standin()
    ns = types.ModuleType("natsort")
    ns.natsorted = _natsorted
    sys.modules["natsort"] = ns
    for n in ["nerfstudio.cameras.camera_utils", "nerfstudio.cameras.camera_paths", "nerfstudio.data.dataparsers",
              "nerfstudio.data.dataparsers.base_dataparser", "nerfstudio.data.dataparsers.nerfstudio_dataparser",
              "nerfstudio.data.utils", "nerfstudio.data.utils.dataparsers_utils", "nerfstudio.data.utils.data_utils",
              "nerfstudio.data.datasets", "nerfstudio.data.datasets.base_dataset", "nerfstudio.plugins",
              "nerfstudio.plugins.registry_dataparser", "nerfstudio.utils.io"]:
        if n not in sys.modules:
            m = base._Stub(n)
            m.__path__ = []
            sys.modules[n] = m
            parent, child = n.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)

    cu = sys.modules["nerfstudio.cameras.camera_utils"]

    def auto_orient_and_center_poses(poses, method="up", center_method="poses"):
        # nerfstudio 1.1.3, the branches FusionSense reaches (normal_nerfstudio.py:74-75: "none" / "none")
        origins = poses[..., :3, 3]
        mean_origin = torch.mean(origins, dim=0)
        if center_method == "poses":
            translation = mean_origin
        elif center_method == "none":
            translation = torch.zeros_like(mean_origin)
        else:
            raise NotImplementedError(center_method)
        assert method == "none", method
        transform = torch.eye(4)
        transform[:3, 3] = -translation
        transform = transform[:3, :]
        return transform @ poses, transform

    cu.auto_orient_and_center_poses = auto_orient_and_center_poses
    cu.get_distortion_params = lambda k1=0.0, k2=0.0, k3=0.0, k4=0.0, p1=0.0, p2=0.0: torch.tensor([k1, k2, k3, k4, p1, p2])
    sys.modules["nerfstudio.cameras"].camera_utils = cu

    cams = sys.modules["nerfstudio.cameras.cameras"]
    cams.Cameras = _Cameras
    cams.CameraType = types.SimpleNamespace(PERSPECTIVE=1, FISHEYE=2, FISHEYE624=3)
    cams.CAMERA_MODEL_TO_TYPE = {"OPENCV": 1}

    sys.modules["nerfstudio.utils.io"].load_from_json = lambda p: json.load(open(p, "r", encoding="utf-8"))
    sys.modules["nerfstudio.utils.rich_utils"].CONSOLE = types.SimpleNamespace(log=lambda *a, **k: None,
                                                                               print=lambda *a, **k: None)

    bd = sys.modules["nerfstudio.data.dataparsers.base_dataparser"]

    @dataclasses.dataclass
    class DataparserOutputs:
        image_filenames: list
        cameras: object
        scene_box: object = None
        mask_filenames: object = None
        dataparser_scale: float = 1.0
        dataparser_transform: object = None
        metadata: dict = dataclasses.field(default_factory=dict)

    bd.DataparserOutputs = DataparserOutputs
    sys.modules["nerfstudio.data.scene_box"].SceneBox = lambda aabb: aabb

    nd = sys.modules["nerfstudio.data.dataparsers.nerfstudio_dataparser"]

    @dataclasses.dataclass
    class NerfstudioDataParserConfig:
        _target: object = None
        data: Path = Path()
        scale_factor: float = 1.0
        downscale_factor: object = None
        scene_scale: float = 1.0
        orientation_method: str = "up"
        center_method: str = "poses"
        auto_scale_poses: bool = True
        eval_mode: str = "fraction"
        train_split_fraction: float = 0.9
        eval_interval: int = 8
        depth_unit_scale_factor: float = 1e-3
        mask_color: object = None
        load_3D_points: bool = False

    @dataclasses.dataclass
    class Nerfstudio:
        config: NerfstudioDataParserConfig
        downscale_factor: object = None

        def _get_fname(self, filepath, data_dir, downsample_folder_prefix="images_"):
            assert self.downscale_factor == 1
            return data_dir / filepath

        def _load_3D_points(self, ply_file_path, transform_matrix, scale_factor):
            import open3d as o3d
            pcd = o3d.io.read_point_cloud(str(ply_file_path))
            if len(pcd.points) == 0:
                return None
            points3D = torch.from_numpy(np.asarray(pcd.points, dtype=np.float32))
            points3D = torch.cat((points3D, torch.ones_like(points3D[..., :1])), -1) @ transform_matrix.T
            points3D *= scale_factor
            points3D_rgb = torch.from_numpy((np.asarray(pcd.colors) * 255).astype(np.uint8))
            return {"points3D_xyz": points3D, "points3D_rgb": points3D_rgb}

    nd.NerfstudioDataParserConfig, nd.Nerfstudio = NerfstudioDataParserConfig, Nerfstudio

    ds = sys.modules["nerfstudio.data.datasets.base_dataset"]

    class InputDataset:
        def __init__(self, dataparser_outputs, scale_factor=1.0):
            import copy
            self._dataparser_outputs = dataparser_outputs
            self.scale_factor = scale_factor
            self.metadata = copy.deepcopy(dataparser_outputs.metadata)
            self.cameras = dataparser_outputs.cameras

    ds.InputDataset = InputDataset

    def get_depth_image_from_path(filepath, height, width, scale_factor, interpolation=None):
        from PIL import Image
        image = np.load(filepath) if Path(filepath).suffix == ".npy" else np.array(Image.open(str(filepath)))
        image = image.astype(np.float64) * scale_factor
        assert image.shape[:2] == (height, width)  # (cv2.resize to the same size is the identity)
        return torch.from_numpy(image[:, :, np.newaxis])

    sys.modules["nerfstudio.data.utils.data_utils"].get_depth_image_from_path = get_depth_image_from_path
    sys.modules["dn_splatter.utils.camera_utils"].euclidean_to_z_depth = None


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not present; goldens can only be regenerated in the authoring container")
    from helpers import write_scene_dir
    _install()
    nn = base._load(os.path.join(REF, "dn_splatter", "data", "normal_nerfstudio.py"), "dn_splatter.data.normal_nerfstudio")
    dd = base._load(os.path.join(REF, "dn_splatter", "data", "dn_dataset.py"), "dn_splatter.data.dn_dataset")
    out = {}
    for variant, seed, normal_format in (("plain", 0, "opencv"), ("ragged", 7, "opengl")):
        with tempfile.TemporaryDirectory() as td:
            root = Path(td)
            write_scene_dir(root, np.random.default_rng(seed), variant=variant)
            cfg = nn.NormalNerfstudioConfig(data=root, output_dir=root, load_3D_points=True, load_pcd_normals=True,
                                            load_normals=True, normal_format=normal_format, load_touches=True)
            parser = nn.NormalNerfstudio(config=cfg, downscale_factor=1)
            o = parser._generate_dataparser_outputs("train")
            md = o.metadata
            p = f"{variant}."
            out[p + "seed"] = np.array(seed)
            out[p + "normal_format_is_opengl"] = np.array(normal_format == "opengl")
            # file order as indices into the NATURALLY sorted stems (strings stay out of the fixture)
            stems = _natsorted([f.stem for f in o.image_filenames])
            out[p + "image_order"] = np.array([stems.index(f.stem) for f in o.image_filenames])
            out[p + "mask_order"] = np.array([stems.index(f.stem) for f in o.mask_filenames])
            out[p + "depth_order"] = np.array([stems.index(Path(f).stem) for f in md["depth_filenames"]])
            out[p + "normal_order"] = np.array([stems.index(Path(f).stem) for f in md["normal_filenames"]])
            c = o.cameras
            out[p + "c2w"] = c.camera_to_worlds.numpy()
            for k in ("fx", "fy", "cx", "cy", "width", "height"):
                out[p + k] = getattr(c, k).numpy().reshape(-1)
            out[p + "dataparser_scale"] = np.array(o.dataparser_scale, dtype=np.float64)
            out[p + "dataparser_transform"] = o.dataparser_transform.numpy()
            out[p + "scale_factor"] = np.array(md["scale_factor"], dtype=np.float64)
            out[p + "transform_matrix"] = md["transform_matrix"].numpy()
            out[p + "points3D_xyz"] = md["points3D_xyz"].numpy()
            out[p + "points3D_rgb"] = md["points3D_rgb"].numpy()
            out[p + "points3D_normals"] = md["points3D_normals"].numpy()
            out[p + "visual_hull"] = md["visual_hull"].numpy()
            out[p + "gel_scale_factor"] = np.array(md["gel_scale_factor"], dtype=np.float64)
            out[p + "n_patches"] = np.array(len(md["touch_patches"]))
            for i, tp in enumerate(md["touch_patches"]):
                for k in ("points_xyz", "points_rgb", "normals", "bbox"):
                    out[p + f"touch{i}.{k}"] = tp[k].numpy()
            ds = dd.GDataset(o)
            for idx in range(len(o.image_filenames)):
                m = ds.get_metadata({"image_idx": idx})
                out[p + f"batch{idx}.sensor_depth"] = m["sensor_depth"].numpy()
                out[p + f"batch{idx}.normal"] = m["normal"].numpy()
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")


if __name__ == "__main__":
    main()
