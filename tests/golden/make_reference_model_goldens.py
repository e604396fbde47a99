#!/usr/bin/env python3
"""Golden vectors made by EXECUTING the reference's own ``DNSplatterModel`` methods in this container:

  get_outputs        (/root/reference/dn_splatter/dn_model.py:469-671) — binary-opacity write, touch-anchor detach,
                     argument preparation of both rasterizer calls, rgb / depth / normal epilogue, per-Gaussian normals
  refinement_after   (:326-451) — densification schedule, split / dup masks, cull, optimizer surgery, opacity reset
  add_touch_patch    (:1156-1247), hull_pruning (:1249-1276), touch_pruning (:1279-1302)

The methods are called unbound on a bare instance.  What they import from gsplat / nerfstudio (not installed) is
supplied by stand-ins: the rasterizer entry points are this repo's CPU oracle (oracle/gsplat_ref.py), the inherited
SplatfactoModel methods (split_gaussians, dup_gaussians, cull_gaussians, dup_in_optim, remove_from_optim, colors,
get_viewmat ...) are restated from nerfstudio 1.1.3 as recalled.  So these goldens pin the
FusionSense-OWNED code of the path — the caller glue and the callbacks — against the repo's restatement of it
(oracle/fusion_ref.py, oracle/splatfacto_ref.py, fusionsense_amd/splatfacto.py, fusionsense_amd/touch.py); the
third-party arithmetic behind the stand-ins stays unpinned, as oracle/gsplat_ref.py's header says.

Runs only where /root/reference exists; writes tests/golden/reference_model.npz (inputs and outputs only).
"""
import dataclasses
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_reference_goldens as base  # noqa: E402
from oracle import gsplat_ref as R  # noqa: E402

OUT = os.path.join(HERE, "reference_model.npz")
NAMES = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]


def _splatfacto_standin():
    m = base._Stub("nerfstudio.models.splatfacto")
    m.__path__ = []

    @dataclasses.dataclass
    class SplatfactoModelConfig:
        pass

    class SplatfactoModel:
        """nerfstudio 1.1.3 SplatfactoModel, the members FusionSense's methods touch (recalled)."""
        training = True
        device = torch.device("cpu")

        means = property(lambda s: s.gauss_params["means"])
        scales = property(lambda s: s.gauss_params["scales"])
        quats = property(lambda s: s.gauss_params["quats"])
        features_dc = property(lambda s: s.gauss_params["features_dc"])
        features_rest = property(lambda s: s.gauss_params["features_rest"])
        opacities = property(lambda s: s.gauss_params["opacities"])
        num_points = property(lambda s: s.gauss_params["means"].shape[0])

        @property
        def colors(self):
            return SH2RGB(self.features_dc) if self.config.sh_degree > 0 else torch.sigmoid(self.features_dc)

        def _get_downscale_factor(self):
            return 1

        def _get_background_color(self):
            return torch.ones(3)

        def get_gaussian_param_groups(self):
            return {name: [self.gauss_params[name]] for name in NAMES + ["normals"] if name in self.gauss_params}

        def split_gaussians(self, split_mask, samps):
            n_splits = int(split_mask.sum().item())
            centered = torch.randn((samps * n_splits, 3))
            self._last_randn = centered.clone()
            scaled = torch.exp(self.scales[split_mask].repeat(samps, 1)) * centered
            quats = self.quats[split_mask] / self.quats[split_mask].norm(dim=-1, keepdim=True)
            rots = R.quat_to_rotmat(quats.repeat(samps, 1))
            rotated = torch.bmm(rots, scaled[..., None]).squeeze()
            new_means = rotated + self.means[split_mask].repeat(samps, 1)
            new_fdc = self.features_dc[split_mask].repeat(samps, 1)
            new_frest = self.features_rest[split_mask].repeat(samps, 1, 1)
            new_opac = self.opacities[split_mask].repeat(samps, 1)
            size_fac = 1.6
            new_scales = torch.log(torch.exp(self.scales[split_mask]) / size_fac).repeat(samps, 1)
            self.scales[split_mask] = torch.log(torch.exp(self.scales[split_mask]) / size_fac)
            new_quats = self.quats[split_mask].repeat(samps, 1)
            out = {"means": new_means, "features_dc": new_fdc, "features_rest": new_frest, "opacities": new_opac,
                   "scales": new_scales, "quats": new_quats}
            for name, param in self.gauss_params.items():
                if name not in out:
                    out[name] = param[split_mask].repeat(samps, 1)
            return out

        def dup_gaussians(self, dup_mask):
            return {name: param[dup_mask] for name, param in self.gauss_params.items()}

        def cull_gaussians(self, extra_cull_mask=None):
            culls = (torch.sigmoid(self.opacities) < self.config.cull_alpha_thresh).squeeze()
            if extra_cull_mask is not None:
                culls = culls | extra_cull_mask
            if self.step > self.config.refine_every * self.config.reset_alpha_every:
                toobigs = (torch.exp(self.scales).max(dim=-1).values > self.config.cull_scale_thresh).squeeze()
                if self.step < self.config.stop_screen_size_at and self.max_2Dsize is not None:
                    toobigs = toobigs | (self.max_2Dsize > self.config.cull_screen_size).squeeze()
                culls = culls | toobigs
            for name, param in self.gauss_params.items():
                self.gauss_params[name] = torch.nn.Parameter(param[~culls])
            return culls

        def remove_from_optim(self, optimizer, deleted_mask, new_params):
            param = optimizer.param_groups[0]["params"][0]
            st = optimizer.state[param]
            del optimizer.state[param]
            if "exp_avg" in st:
                st["exp_avg"] = st["exp_avg"][~deleted_mask]
                st["exp_avg_sq"] = st["exp_avg_sq"][~deleted_mask]
            del optimizer.param_groups[0]["params"][0]
            optimizer.param_groups[0]["params"] = new_params
            optimizer.state[new_params[0]] = st

        def dup_in_optim(self, optimizer, dup_mask, new_params, n=2):
            param = optimizer.param_groups[0]["params"][0]
            st = optimizer.state[param]
            if "exp_avg" in st:
                rep = (n,) + tuple(1 for _ in range(st["exp_avg"].dim() - 1))
                st["exp_avg"] = torch.cat([st["exp_avg"], torch.zeros_like(st["exp_avg"][dup_mask.squeeze()]).repeat(*rep)], dim=0)
                st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"], torch.zeros_like(st["exp_avg_sq"][dup_mask.squeeze()]).repeat(*rep)], dim=0)
            del optimizer.state[param]
            optimizer.state[new_params[0]] = st
            optimizer.param_groups[0]["params"] = new_params

    C0 = 0.28209479177387814

    def SH2RGB(sh):
        return sh * C0 + 0.5

    def RGB2SH(rgb):
        return (rgb - 0.5) / C0

    m.SplatfactoModel, m.SplatfactoModelConfig = SplatfactoModel, SplatfactoModelConfig
    m.RGB2SH, m.get_viewmat = RGB2SH, R.get_viewmat
    return m


class FakeCameras:
    """What get_outputs reads from nerfstudio's Cameras (one camera)."""

    def __init__(self, c2w, fx, fy, cx, cy, W, H):
        self.camera_to_worlds = c2w[None]
        self._k = torch.tensor([[[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]]])
        self.width, self.height = torch.tensor([[W]]), torch.tensor([[H]])
        self.shape = (1,)
        self.metadata = None

    def rescale_output_resolution(self, f):
        pass

    def get_intrinsics_matrices(self):
        return self._k


class Optimizers:
    def __init__(self, params):
        self.optimizers = {k: torch.optim.Adam([v], lr=1e-3, eps=1e-15) for k, v in params.items()}


def make_model(model_mod, params, step, cfg_over=None):
    cfg = types.SimpleNamespace(
        use_binary_opacities=True, binary_opacities_threshold=0.9, warmup_length=500, reset_alpha_every=30,
        refine_every=100, sh_degree=3, sh_degree_interval=1000, rasterize_mode="classic", predict_normals=True,
        stop_split_at=10000, densify_grad_thresh=0.0008, densify_size_thresh=0.01, stop_screen_size_at=4000,
        split_screen_size=0.05, n_split_samples=2, cull_alpha_thresh=0.1, cull_scale_thresh=0.5, cull_screen_size=0.15,
        continue_cull_post_densification=True, add_touch_at=1000, use_sdf_loss=False)
    for k, v in (cfg_over or {}).items():
        setattr(cfg, k, v)
    self = object.__new__(model_mod.DNSplatterModel)
    self.config, self.step = cfg, step
    self.gauss_params = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
    self.crop_box, self.add_mask, self.added_count = None, None, 0
    self.camera_optimizer = types.SimpleNamespace(apply_to_camera=lambda cam: cam.camera_to_worlds)
    self.kwargs = {"metadata": {}}
    self.xys_grad_norm = self.vis_counts = self.max_2Dsize = None
    self.num_train_data = 9
    return self


def save(out, prefix, d):
    for k, v in d.items():
        if v is None:
            continue
        out[f"{prefix}.{k}"] = v.detach().clone().numpy() if torch.is_tensor(v) else np.array(v)  # (a copy: later in-place edits must not reach it)


def load_reference_model():
    """The reference's dn_model module, loaded over the stand-ins (rasterizer = oracle/gsplat_ref.py)."""
    if not os.path.isdir(base.REF):
        raise SystemExit("reference not present")
    base._install_stubs()
    for name in [n for n in sys.modules if n == "sklearn" or n.startswith("sklearn.")]:
        if isinstance(sys.modules[name], base._Stub):
            del sys.modules[name]  # (sklearn is installed; torch's optimizers probe sys.modules entries for a __spec__)
    import importlib.machinery
    for name, mod in list(sys.modules.items()):
        if isinstance(mod, base._Stub) and getattr(mod, "__spec__", None) is None:
            mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
    sys.modules["nerfstudio.models.splatfacto"] = _splatfacto_standin()
    sys.modules["nerfstudio.models.splatfacto"].__spec__ = importlib.machinery.ModuleSpec("nerfstudio.models.splatfacto", None)
    sys.modules["nerfstudio.models"].splatfacto = sys.modules["nerfstudio.models.splatfacto"]
    torch.Tensor.cuda = lambda self, *a, **k: self  # (the reference calls .cuda() on a few tensors)
    base._load(os.path.join(base.REF, "dn_splatter", "losses.py"), "dn_splatter.losses")
    mm = base._load(os.path.join(base.REF, "dn_splatter", "dn_model.py"), "dn_splatter.dn_model")
    mm.rasterization, mm.rasterize_gaussians, mm.quat_to_rotmat = R.rasterization, R.rasterize_gaussians, R.quat_to_rotmat
    mm.num_sh_bases = lambda d: (d + 1) ** 2
    mm.Cameras = FakeCameras
    mm.get_viewmat = R.get_viewmat
    mm.CONSOLE = types.SimpleNamespace(log=lambda *a, **k: None, print=lambda *a, **k: None)

    def knn_exact(x, y, k):  # dn_splatter.utils.knn.knn_sk(x, y, k): for every row of y the k nearest rows of x
        return torch.cdist(y.double(), x.double()).topk(k, largest=False).indices
    mm.knn_sk = knn_exact
    return mm


def main():
    mm = load_reference_model()
    from fusionsense_amd import scenes
    out = {}

    # ---------------- get_outputs: plain / binary-opacity write + touch anchors ----------------
    params, cam = scenes.cube_scene(260, seed=21)
    W, H = 56, 40
    fcam = FakeCameras(cam.c2w, 60.0, 58.0, 27.5, 20.5, W, H)
    save(out, "go.in", params)
    out["go.c2w"], out["go.intr"] = cam.c2w.numpy(), np.array([60.0, 58.0, 27.5, 20.5, W, H])
    g = torch.Generator().manual_seed(3)
    wts = {k: torch.rand(H, W, c, generator=g) for k, c in (("rgb", 3), ("depth", 1), ("normal", 3))}
    save(out, "go.w", wts)
    for case, step, with_anchor in (("plain", 100, False), ("binary_anchor", 3300, True)):
        m = make_model(mm, params, step)
        if with_anchor:
            am = torch.zeros(260, dtype=torch.bool)
            am[-40:] = True
            m.add_mask = am
            out[f"go.{case}.add_mask"] = am.numpy()
        o = m.get_outputs(fcam)
        loss = sum((o[k] * wts[k]).mean() for k in ("rgb", "depth", "normal"))
        loss.backward()
        save(out, f"go.{case}.out", {k: o[k] for k in ("rgb", "depth", "normal", "accumulation")})
        save(out, f"go.{case}.grad", {k: (m.gauss_params[k].grad if m.gauss_params[k].grad is not None
                                          else torch.zeros_like(m.gauss_params[k])) for k in NAMES})
        out[f"go.{case}.opacities_after"] = m.gauss_params["opacities"].detach().numpy()
        out[f"go.{case}.normals_world"] = m.gauss_params["normals"].detach().numpy()
        out[f"go.{case}.step"] = np.array(step)
        out[f"go.{case}.sh_degree_to_use"] = np.array(min(step // 1000, 3))

    # ---------------- refinement_after at several points of the schedule ----------------
    n = 400
    g = torch.Generator().manual_seed(8)
    p2, _ = scenes.cube_scene(n, seed=5)
    p2["scales"] = torch.log(0.002 + 0.03 * torch.rand(n, 3, generator=g))
    p2["opacities"] = 2 * torch.randn(n, 1, generator=g)
    p2["normals"] = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    stats = dict(xys_grad_norm=torch.rand(n, generator=g) * 0.01, vis_counts=1 + torch.randint(0, 5, (n,), generator=g).float(),
                 max_2Dsize=torch.rand(n, generator=g) * 0.2)
    save(out, "ra.in", p2)
    save(out, "ra.stats", stats)
    for step in (400, 700, 3100, 3500, 10000):
        m = make_model(mm, p2, step)
        for k, v in stats.items():
            setattr(m, k, v.clone())
        m.last_size = (720, 1280)
        if step == 3500:
            am = torch.zeros(n, dtype=torch.bool)
            am[:50] = True
            m.add_mask = am
        opt = Optimizers(m.gauss_params)
        for k, p in m.gauss_params.items():  # warm Adam moments
            p.grad = torch.sin(p.detach() * 3.0)
            opt.optimizers[k].step()
        pre = {k: v.detach().clone() for k, v in m.gauss_params.items()}
        pre_m = {k: opt.optimizers[k].state[m.gauss_params[k]]["exp_avg"].clone() for k in m.gauss_params}
        pre_v = {k: opt.optimizers[k].state[m.gauss_params[k]]["exp_avg_sq"].clone() for k in m.gauss_params}
        torch.manual_seed(1234)
        m._last_randn = torch.zeros(0, 3)
        m.refinement_after(opt, step)
        tag = f"ra.{step}"
        save(out, tag + ".pre", pre)
        save(out, tag + ".pre_m", pre_m)
        save(out, tag + ".pre_v", pre_v)
        save(out, tag + ".post", {k: v.detach() for k, v in m.gauss_params.items()})
        save(out, tag + ".post_m", {k: opt.optimizers[k].state[m.gauss_params[k]]["exp_avg"] for k in m.gauss_params})
        save(out, tag + ".post_v", {k: opt.optimizers[k].state[m.gauss_params[k]]["exp_avg_sq"] for k in m.gauss_params})
        out[tag + ".randn"] = m._last_randn.numpy()
        out[tag + ".add_mask_after"] = (m.add_mask.numpy() if m.add_mask is not None else np.zeros(0, dtype=bool))
        out[tag + ".stats_cleared"] = np.array(m.xys_grad_norm is None and m.max_2Dsize is None)

    # ---------------- add_touch_patch, touch_pruning, hull_pruning ----------------
    g = torch.Generator().manual_seed(12)
    n = 500
    p3, _ = scenes.cube_scene(n, seed=9)
    p3["means"] = p3["means"] * 0.4
    p3["opacities"] = 1.0 + torch.randn(n, 1, generator=g)
    p3["normals"] = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    patches = scenes.touch_patches_on_blob(2, 60, radius=0.2, seed=1, extent=(0.05, 0.04), depth=0.08)
    patches.append({"points_xyz": torch.zeros(0, 3), "points_rgb": torch.zeros(0, 3), "normals": torch.zeros(0, 3),
                    "bbox": patches[0]["bbox"]})
    hull = 0.1 * torch.nn.functional.normalize(torch.randn(300, 3, generator=g), dim=-1)
    m = make_model(mm, p3, 1000)
    m.kwargs["metadata"] = {"touch_patches": patches, "gel_scale_factor": 6.34e-5, "visual_hull": hull, "scale_factor": 1.3}
    opt = Optimizers(m.gauss_params)
    for k, p in m.gauss_params.items():
        p.grad = torch.cos(p.detach() * 2.0)
        opt.optimizers[k].step()
    save(out, "tp.in", {k: v.detach() for k, v in m.gauss_params.items()})
    save(out, "tp.in_m", {k: opt.optimizers[k].state[m.gauss_params[k]]["exp_avg"] for k in m.gauss_params})
    for i, pa in enumerate(patches):
        save(out, f"tp.patch{i}", pa)
    out["tp.hull"] = hull.numpy()
    m.add_touch_patch(opt, 1000)
    save(out, "tp.added", {k: v.detach() for k, v in m.gauss_params.items()})
    save(out, "tp.added_m", {k: opt.optimizers[k].state[m.gauss_params[k]]["exp_avg"] for k in m.gauss_params})
    out["tp.add_mask"] = m.add_mask.numpy()
    m.step = 1100
    with torch.no_grad():
        m.gauss_params["means"][:7] = patches[0]["points_xyz"][:7] - 0.01 * patches[0]["normals"][:7]  # inside the box
    save(out, "tp.before_touch_prune", {"means": m.gauss_params["means"].detach()})
    m.touch_pruning(opt, 1100)
    save(out, "tp.touch_pruned", {k: v.detach() for k, v in m.gauss_params.items()})
    out["tp.add_mask_touch_pruned"] = m.add_mask.numpy()
    m.hull_pruning(opt, 1100)
    save(out, "tp.hull_pruned", {k: v.detach() for k, v in m.gauss_params.items()})
    out["tp.add_mask_hull_pruned"] = m.add_mask.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays;", {k: out[k].shape for k in out if k.endswith(".means")})


if __name__ == "__main__":
    main()
