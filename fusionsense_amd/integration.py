"""Optional second step of the integration: ``DNSplatterModel.get_outputs`` itself on the fused node.

``gsplat_shim.install()`` keeps FusionSense's model file byte-for-byte and swaps the rasterizer underneath its two
calls; that route is bounded by the reference's own call structure (two compositing passes, ~60 small torch launches
of glue per iteration).  This module offers the same method — /root/reference/dn_splatter/dn_model.py:469-671, same
inputs, same returned dictionary, same side effects on the model object — computed by the fused node of
``fusionsense_amd/fused.py`` (one compositing walk for RGB+ED and the normal plane, no torch glue), still on torch's
autograd tape and under nerfstudio's own optimizers and callbacks:

    import dn_splatter.dn_model as m
    from fusionsense_amd import integration
    integration.patch(m.DNSplatterModel)          # get_outputs -> get_outputs_fused (the original stays as
                                                  # _get_outputs_reference and is used for what the node does not cover)

``integration.patch_all(m.DNSplatterModel)`` goes one step further and routes the REST of the model's training-loop
surface to the HIP path as well — same method names, same arguments, still driven by nerfstudio's Trainer, its
callbacks and its ``Optimizers`` (round 3):

    get_loss_dict      dn_model.py:673-925   -> ops._FusionLoss over a cached FrameBatch (one autograd node: every switch
                                                reachable from scripts/train.py — incl. normals from depth — and
                                                EdgeAwareTV, the cosine term, the L1 / LogL1 / MSE depth losses); the
                                                HuberL1, the sparse-opacity term, the scale regularisation and the SDF
                                                term (sdf.py) are small torch terms on the node's outputs; the original
                                                stays as ``_get_loss_dict_reference`` for batches / shapes the node
                                                does not take
    get_metrics_dict   dn_model.py:927-1003  -> the per-iteration PSNR / SSIM / depth metrics on the device, one transfer
    after_train        (nerfstudio, A.2)     -> nothing to do: the statistics were applied by the node's backward
    refinement_after   dn_model.py:326-451   -> DensifyStrategy.refinement_after (HIP row compaction, split sampling)
    add_touch_patch    dn_model.py:1156-1247 -> DensifyStrategy.add_touch_patch
    hull_pruning       dn_model.py:1249-1276 -> DensifyStrategy.hull_pruning
    touch_pruning      dn_model.py:1279-1302 -> DensifyStrategy.touch_pruning
    fuse_optimizers(optimizers.optimizers)   -> the Gaussian groups' torch.optim.Adam objects keep their identity, state
                                                layout and schedulers; their ``step()`` calls are gathered and ONE
                                                fsgs_adam_step launch runs when the last live group has stepped

Side effects reproduced (they are what after_train / refinement_after / get_loss_dict read, dn_model.py:592-600,
607, 634, 660-663): ``self.xys`` (an object whose ``.absgrad`` [1,N,2] is filled by the backward), ``self.radii``,
``self.depths``, ``self.conics``, ``self.num_tiles_hit``, ``self.vis_indices``, ``self.last_size``,
``self.gauss_params["normals"]``, ``self.camera``, ``self.camera_idx``.
"""
from __future__ import annotations

from typing import Dict

import torch

from .scenes import Camera


def _binary_threshold(cfg, step: int):
    """dn_model.py:492-503: is the binary-opacity write performed at this step?"""
    if getattr(cfg, "use_binary_opacities", False) and step > cfg.warmup_length:
        skip_steps = cfg.reset_alpha_every * cfg.refine_every
        if not step % skip_steps == 0 and step % skip_steps not in range(1, 200 + 1):
            return float(cfg.binary_opacities_threshold)
    return None


def get_outputs_fused(self, camera) -> Dict[str, torch.Tensor]:
    """Drop-in for ``DNSplatterModel.get_outputs(camera)``."""
    from .fused import render_fusionsense_fused
    if not hasattr(camera, "camera_to_worlds"):
        print("Called get_outputs with not a camera")
        return {}
    cfg = self.config
    if cfg.sh_degree <= 0 or not cfg.predict_normals or cfg.rasterize_mode != "classic":
        return self._get_outputs_reference(camera)  # (sigmoid colours / no normals / antialiased: the op-by-op route)
    if self.training:
        assert camera.shape[0] == 1, "Only one camera at a time"
        c2w = self.camera_optimizer.apply_to_camera(camera)
    else:
        c2w = camera.camera_to_worlds
    if torch.is_tensor(c2w) and c2w.requires_grad:
        return self._get_outputs_reference(camera)  # (camera optimisation is off in FusionSense, dn_model.py:128-130)
    scale = self._get_downscale_factor()
    camera.rescale_output_resolution(1 / scale)
    K = camera.get_intrinsics_matrices()[0].detach().cpu()
    W, H = int(camera.width.item()), int(camera.height.item())
    camera.rescale_output_resolution(scale)
    self.last_size = (H, W)
    cam = Camera(c2w.reshape(-1, 3, 4)[0].detach().cpu().float(), float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
                 float(K[1, 2]), W, H)
    self.__dict__["_fsgs_intrinsics"] = (cam.fx, cam.fy, cam.cx, cam.cy)
    dev = self.gauss_params["means"].device
    sh_degree_to_use = min(self.step // cfg.sh_degree_interval, cfg.sh_degree)
    background = self._get_background_color().to(device=dev, dtype=torch.float32)
    crop_box = self.crop_box if (self.crop_box is not None and not self.training) else None
    stats = None
    if self.training and getattr(type(self), "_fsgs_patched_all", False) and self.step < cfg.stop_split_at \
            and torch.is_grad_enabled():
        # patch_all: the node's backward applies after_train itself (nerfstudio creates the buffers the same way on
        # first use: zeros / ones / zeros)
        N = self.gauss_params["means"].shape[0]
        if getattr(self, "xys_grad_norm", None) is None or self.xys_grad_norm.shape[0] != N:
            self.xys_grad_norm = torch.zeros(N, device=dev, dtype=torch.float32)
            self.vis_counts = torch.ones(N, device=dev, dtype=torch.float32)
        if getattr(self, "max_2Dsize", None) is None or self.max_2Dsize.shape[0] != N:
            self.max_2Dsize = torch.zeros(N, device=dev, dtype=torch.float32)
        stats = {"xys_grad_norm": self.xys_grad_norm, "vis_counts": self.vis_counts, "max_2Dsize": self.max_2Dsize,
                 "inv_max_hw": 1.0 / float(max(H, W))}
    out = render_fusionsense_fused(self.gauss_params, cam, sh_degree=sh_degree_to_use, background=background, device=dev,
                                   add_mask=self.add_mask, crop_box=crop_box, training=bool(self.training),
                                   binary_threshold=_binary_threshold(cfg, self.step), stats_out=stats)
    if "info" not in out:  # the crop left nothing: get_empty_outputs
        return out
    info = out["info"]
    self.xys = info                      # .absgrad [1,N,2] after backward, as after_train reads it
    self.radii = info.radii[0]
    self.depths, self.conics, self.num_tiles_hit = info.depths, info.conics, info.tiles_per_gauss
    self.vis_indices = torch.where(self.radii > 0)[0]
    if crop_box is None:
        self.gauss_params["normals"] = out["normals_world"]
    md = getattr(camera, "metadata", None)
    if md is not None and "cam_idx" in md:
        self.camera_idx = md["cam_idx"]
    self.camera = camera
    return {"rgb": out["rgb"], "depth": out["depth"], "normal": out["normal"], "accumulation": out["accumulation"],
            "background": background}


def patch(model_cls) -> None:
    """Route ``model_cls.get_outputs`` through the fused node (idempotent)."""
    if getattr(model_cls, "_fsgs_patched", False):
        return
    model_cls._get_outputs_reference = model_cls.get_outputs
    model_cls.get_outputs = get_outputs_fused
    model_cls._fsgs_patched = True


# ---------------------------------------------------------------------------------------------------------------------
# patch_all: get_loss_dict, after_train and the refinement / touch / hull callbacks on the HIP path (round 3)
# ---------------------------------------------------------------------------------------------------------------------
GAUSS_GROUPS = ("means", "scales", "quats", "features_dc", "features_rest", "opacities")


def _loss_config(cfg):
    """(losses.LossConfig, extras) from the model's DNSplatterModelConfig, or None when a switch is set that neither the
    fused loss node nor the torch terms on top of it evaluate (the caller then runs the reference's own get_loss_dict).
    ``extras`` names the terms that stay small
    torch expressions on the node's outputs (``_extra_terms``): "huber" — the HuberL1 depth loss, whose threshold is a
    maximum over the frame's valid pixels (losses.py:217-238); "sparse" — the entropy of the visible Gaussians'
    opacities (:821-837); "scale_reg" — splatfacto's PhysGaussian scale regularisation; "sdf" — the SuGaR-style SDF term
    (:838-882) over random surface samples (fusionsense_amd/sdf.py; every apply_sdf_loss_iters-th step after
    apply_sdf_loss_after_iters, on the neighbour table ``self._knn`` the reference's own recompute_knn callback keeps)."""
    from .losses import LossConfig

    def name_of(v):  # enum member / string -> lower-case name
        return str(getattr(v, "name", v)).lower()
    g = lambda k, d=None: getattr(cfg, k, d)  # noqa: E731
    extras = set()
    if g("use_sdf_loss", False):
        extras.add("sdf")  # (round 5: fusionsense_amd/sdf.py, pinned by tests/golden/reference_sdf.npz)
    if g("use_scale_regularization", False):
        extras.add("scale_reg")
    if g("use_sparse_loss", False):
        extras.add("sparse")
    types = {"edgeawarelogl1": "EdgeAwareLogL1", "l1": "L1", "logl1": "LogL1", "mse": "MSE", "huberl1": "HuberL1"}
    dtype = types.get(name_of(g("depth_loss_type", "EdgeAwareLogL1")).split(".")[-1])
    if g("use_depth_loss", False) and dtype is None:
        return None
    lam_sensor, lam_mono = float(g("sensor_depth_lambda", 0.0)), float(g("mono_depth_lambda", 0.0))
    if g("use_depth_loss", False) and dtype == "HuberL1":
        extras.add("huber")  # (the node keeps the smoothness term; the two data terms are added in torch)
        dtype, lam_sensor, lam_mono = "EdgeAwareLogL1", 0.0, 0.0
    smooth = {"tv": "TV", "edgeawaretv": "EdgeAwareTV"}.get(name_of(g("smooth_loss_type", "TV")).split(".")[-1])
    if g("use_depth_smooth_loss", False) and smooth is None:
        return None
    supervision = str(g("normal_supervision", "mono"))
    if g("use_normal_loss", False) and supervision not in ("mono", "depth"):
        return None
    return LossConfig(normal_supervision=supervision, smooth_loss_type=smooth or "TV",
                      use_normal_cosine_loss=bool(g("use_normal_cosine_loss", False)),
                      depth_loss_type=dtype or "EdgeAwareLogL1",
                      ssim_lambda=float(g("ssim_lambda", 0.2)), use_depth_loss=bool(g("use_depth_loss", False)),
                      sensor_depth_lambda=lam_sensor, mono_depth_lambda=lam_mono,
                      depth_tolerance=float(g("depth_tolerance", 0.1)),
                      use_depth_smooth_loss=bool(g("use_depth_smooth_loss", False)),
                      smooth_loss_lambda=float(g("smooth_loss_lambda", 0.1)),
                      use_normal_loss=bool(g("use_normal_loss", False)),
                      use_normal_tv_loss=bool(g("use_normal_tv_loss", False)),
                      normal_lambda=float(g("normal_lambda", 0.1)), two_d_gaussians=bool(g("two_d_gaussians", False)),
                      touch_normal_loss_lambda=1.0), frozenset(extras)


def _extra_terms(self, outputs, batch, extras):
    """The loss terms that stay torch code (see ``_loss_config``) -> (term added to main_loss or None, scale_reg or None).
    A handful of small launches on the steps that have them; autograd carries their gradients."""
    cfg = self.config
    dev = outputs["rgb"].device
    main = None
    if "huber" in extras:
        # dn_model.py:702-709, 721-750 with losses.py:217-238 as called: the valid pixels as 1-D tensors, the threshold a
        # fifth of the frame's largest error (mask = gt != 0 is all true for them)
        depth_out = outputs["depth"]
        mask = batch["mask"].to(dev) if "mask" in batch else None
        if mask is not None:
            depth_out = depth_out * mask

        def huber(pred, gt):
            l1 = torch.abs(pred - gt)
            d = 0.2 * torch.max(l1)
            return torch.where(l1 < d, ((pred - gt) ** 2 + d ** 2) / (2 * d), l1).mean()
        for key, lam, tol in (("sensor_depth", float(getattr(cfg, "sensor_depth_lambda", 0.0)),
                               float(getattr(cfg, "depth_tolerance", 0.1))),
                              ("mono_depth", float(getattr(cfg, "mono_depth_lambda", 0.0)), 0.0)):
            if key in batch and lam > 0.0:
                gt = batch[key].to(dev)
                if mask is not None:
                    gt = gt * mask
                valid = gt > tol
                term = lam * huber(depth_out[valid], gt[valid].float())
                main = term if main is None else main + term
    if "sparse" in extras:
        # dn_model.py:821-837: every sparse_loss_steps-th step, not on an opacity reset's step nor in the 100 steps behind it
        step = int(self.step)
        period = int(cfg.reset_alpha_every) * int(cfg.refine_every)
        if step % int(cfg.sparse_loss_steps) == 0 and step % period != 0 and not (1 <= step % period <= 100):
            o = torch.sigmoid(self.gauss_params["opacities"][self.vis_indices])
            term = float(cfg.sparse_lambda) * (-o * torch.log(o + 1e-10) - (1 - o) * torch.log(1 - o + 1e-10)).mean()
            main = term if main is None else main + term
    if "sdf" in extras:
        from . import sdf
        if sdf.sdf_term_due(cfg, int(self.step)):
            P = self.gauss_params
            cam = self.camera
            c2w = cam.camera_to_worlds
            c2w = (c2w.squeeze(0) if c2w.dim() == 3 else c2w).to(dev)
            term = float(cfg.sdf_loss_lambda) * sdf.sdf_loss(
                P["means"], P["scales"], P["quats"], P["opacities"], self._knn.to(dev), self.vis_indices,
                getattr(self, "add_mask", None), outputs["depth"], c2w, float(cam.fx.item()), float(cam.cx.item()),
                float(cam.cy.item()), int(cam.width.item()), int(cam.height.item()),
                batch["mask"].to(dev) if "mask" in batch else None, int(cfg.num_sdf_samples))
            main = term if main is None else main + term
    scale_reg = None
    if "scale_reg" in extras and int(self.step) % 10 == 0:
        # nerfstudio 1.1.3 splatfacto.get_loss_dict (un-vendored; published algorithm, PhysGaussian): a tenth of the mean
        # excess of the largest over the smallest scale above max_gauss_ratio
        e = torch.exp(self.gauss_params["scales"])
        ratio = e.amax(dim=-1) / e.amin(dim=-1)
        cap = float(getattr(cfg, "max_gauss_ratio", 10.0))
        scale_reg = 0.1 * (torch.maximum(ratio, torch.tensor(cap, device=ratio.device)) - cap).mean()
    return main, scale_reg


class SupervisionCache:
    """Prepared supervision (``losses.FrameBatch``) of the views ``get_loss_dict`` has seen, resident on the device.

    The reference rebuilds its ground truth from the batch on every call (dn_model.py:683-735); its datamanager hands
    out a ``deepcopy`` of ``cached_train[image_idx]`` / ``cached_eval[image_idx]`` each time (dn_datamanager.py:103,
    161), i.e. the same DATA under new tensors.  So an entry is keyed on **(split, image_idx)** — nerfstudio's pipeline
    also calls ``get_loss_dict`` with EVAL batches (``get_eval_loss_dict``), whose indices count the eval dataset and
    would otherwise collide with the train view of the same number — and a hit is verified:

    * shapes and dtypes of every supervision tensor must match the entry's;
    * the very tensors the entry was prepared from (same ``data_ptr`` and version) are a hit at once;
    * otherwise a sampled fingerprint of the batch (~4 k strided elements of each tensor, bit patterns) is compared
      with the entry's ON THE DEVICE — three small launches, no host wait in the step — and the flag is read when the next
      call finds it finished: a batch that carried other data under a cached key raises ``RuntimeError`` there (loudly, one call
      late) and the entry is dropped.  ``FSGS_FRAME_CACHE_VERIFY=1`` compares in full and synchronously (tests).

    A batch without ``image_idx`` is prepared uncached.  Least recently used entries leave once the cache holds more
    than ``FSGS_FRAME_CACHE_MB`` (default 8192; ~36 B per pixel and view).  ``FSGS_FRAME_CACHE=0`` disables caching."""
    KEYS = ("image", "sensor_depth", "mono_depth", "normal", "mask")

    def __init__(self, budget_bytes=None):
        import os
        from collections import OrderedDict
        self.entries = OrderedDict()
        self.enabled = os.environ.get("FSGS_FRAME_CACHE", "1") != "0"
        self.verify_full = os.environ.get("FSGS_FRAME_CACHE_VERIFY", "0") == "1"
        self.budget = int(float(os.environ.get("FSGS_FRAME_CACHE_MB", "8192")) * (1 << 20)) if budget_bytes is None \
            else int(budget_bytes)
        self.bytes = 0
        self.hits = self.misses = self.evictions = 0
        self._checks = []  # (key, device flag, event) of fingerprint comparisons not read yet

    def __len__(self):
        return len(self.entries)

    @staticmethod
    def _signature(batch):
        return tuple((k, tuple(batch[k].shape), batch[k].dtype) for k in SupervisionCache.KEYS if k in batch)

    @staticmethod
    def _identity(batch):
        return tuple((batch[k].data_ptr(), batch[k]._version) for k in SupervisionCache.KEYS if k in batch)

    @staticmethod
    def _fingerprint(batch, device):
        """A strided sample (~4 k elements per tensor, bit patterns) of the batch's supervision tensors as ONE device
        vector: one gather-free view per tensor, one concatenation — cheap enough to take on every call that hands over
        re-created tensors (the reference's datamanager does, every step)."""
        parts = []
        for k in SupervisionCache.KEYS:
            if k in batch:
                t = batch[k].to(device).reshape(-1)
                step = max(1, t.numel() // 4096)
                t = t[::step]
                if t.dtype != torch.float32:
                    t = t.to(torch.float32)
                parts.append(t.view(torch.int32))
        return torch.cat(parts)

    @staticmethod
    def _nbytes(fb):
        n = 0
        for t in (fb.image, fb.sensor_depth, fb.normal, fb.mask, fb.mono_depth):
            if torch.is_tensor(t):
                n += t.numel() * t.element_size()
        return n

    def _harvest(self, block=False):
        keep = []
        for key, flag, ev in self._checks:
            if block or ev is None or ev.query():
                if bool(flag.item()):
                    self._drop(key)
                    self._checks = [c for c in self._checks if c[0] != key]
                    raise RuntimeError(
                        f"get_loss_dict: the batch of {'train' if key[0] else 'eval'} image_idx {key[1]} carried other "
                        "data than the supervision cached for that view (SupervisionCache; the entry has been dropped — "
                        "FSGS_FRAME_CACHE=0 prepares every batch afresh)")
            else:
                keep.append((key, flag, ev))
        self._checks = keep

    def _drop(self, key):
        e = self.entries.pop(key, None)
        if e is not None:
            self.bytes -= e["bytes"]

    def clear(self):
        self.entries.clear()
        self._checks = []
        self.bytes = 0

    def get(self, training, batch, cfg, device, height, width):
        from .losses import prepare_batch
        sub = {k: batch[k] for k in self.KEYS if k in batch}
        if not self.enabled or "image_idx" not in batch:
            return prepare_batch(sub, cfg, device)
        self._harvest()
        key = (bool(training), int(batch["image_idx"]))
        sig = self._signature(sub)
        e = self.entries.get(key)
        if e is not None and (e["sig"] != sig or e["fb"].height != height or e["fb"].width != width
                              or e["tol"] != float(cfg.depth_tolerance)):
            self._drop(key)
            e = None
        if e is not None:
            ident = self._identity(sub)
            if ident != e["ident"]:
                if self.verify_full:
                    fresh = prepare_batch(sub, cfg, device)
                    same = all((a is None and b is None) or (a is not None and b is not None and torch.equal(a, b))
                               for a, b in ((fresh.image, e["fb"].image), (fresh.sensor_depth, e["fb"].sensor_depth),
                                            (fresh.normal, e["fb"].normal), (fresh.mask, e["fb"].mask),
                                            (fresh.mono_depth, e["fb"].mono_depth)))
                    if not same:
                        self._drop(key)
                        raise RuntimeError(f"get_loss_dict: batch of {'train' if key[0] else 'eval'} image_idx {key[1]} "
                                           "differs from the cached supervision of that view")
                else:
                    fp = self._fingerprint(sub, device)
                    flag = (fp != e["fp"]).any() if fp.shape == e["fp"].shape else torch.ones((), dtype=torch.bool, device=fp.device)
                    ev = None
                    if flag.is_cuda:
                        ev = torch.cuda.Event()
                        ev.record()
                    self._checks.append((key, flag, ev))
            self.entries.move_to_end(key)
            self.hits += 1
            return e["fb"]
        fb = prepare_batch(sub, cfg, device)
        nb = self._nbytes(fb)
        self.misses += 1
        if nb <= self.budget:
            self.entries[key] = dict(fb=fb, sig=sig, ident=self._identity(sub), fp=self._fingerprint(sub, device), bytes=nb,
                                     tol=float(cfg.depth_tolerance))
            self.bytes += nb
            while self.bytes > self.budget and len(self.entries) > 1:
                old = next(iter(self.entries))
                self._drop(old)
                self.evictions += 1
        return fb


def _intrinsics_of(model):
    """(fx, fy, cx, cy) of the camera of the model's last get_outputs — what normal_supervision == "depth" reads from
    ``self.camera`` (dn_model.py:775-786)."""
    cam = getattr(model, "camera", None)
    try:
        return (float(cam.fx.item()), float(cam.fy.item()), float(cam.cx.item()), float(cam.cy.item()))
    except (AttributeError, ValueError, RuntimeError):
        return model.__dict__.get("_fsgs_intrinsics")


def get_loss_dict_fused(self, outputs, batch, metrics_dict=None) -> Dict[str, torch.Tensor]:
    """Drop-in for ``DNSplatterModel.get_loss_dict`` (dn_model.py:673-925): ``main_loss`` is one HIP autograd node over
    the view's supervision, prepared once per (split, ``image_idx``) and kept on the device (``self._fsgs_frames``, a
    :class:`SupervisionCache`: train and eval batches never share an entry, hits are verified, bounded).  What the
    node does not cover goes to the reference's own method.  Not reproduced: the ``log_images/<step>.jpg`` debug dump of
    every 100th step (:903-921) and the in-place masking of ``outputs["normal"]`` / ``batch["normal"]`` (:712-715),
    which nothing reads after the loss."""
    from . import ops
    cfg, extras = _loss_config(self.config) or (None, frozenset())
    rgb = outputs.get("rgb")
    covered = (cfg is not None and torch.is_tensor(rgb) and rgb.is_cuda and "normal" in outputs and "depth" in outputs
               and not (cfg.use_normal_loss and cfg.normal_supervision == "mono" and "normal" not in batch)
               and not (cfg.use_normal_loss and cfg.normal_supervision == "depth" and _intrinsics_of(self) is None)
               and int(getattr(self.config, "num_downscales", 0)) == 0)
    if not covered:
        return self._get_loss_dict_reference(outputs, batch, metrics_dict)
    frames = self.__dict__.get("_fsgs_frames")
    if frames is None:
        frames = self.__dict__["_fsgs_frames"] = SupervisionCache()
    fb = frames.get(bool(self.training), batch, cfg, rgb.device, rgb.shape[0], rgb.shape[1])
    touch_idx = touch_normals = None
    add_mask = getattr(self, "add_mask", None)
    if add_mask is not None:
        touch_idx = torch.nonzero(add_mask.to(rgb.device), as_tuple=False).reshape(-1).contiguous()  # int64 rows
        tn = self.__dict__.get("_fsgs_touch_normals")
        if tn is None or tn.shape[0] != touch_idx.numel():
            tn = torch.cat([tp["normals"].to(rgb.device) for tp in self.kwargs["metadata"]["touch_patches"]], dim=0)
            self.__dict__["_fsgs_touch_normals"] = tn = tn.to(torch.float32).contiguous()
        touch_normals = tn
    main_loss = ops._FusionLoss.apply(rgb, outputs["depth"], outputs["normal"], self.gauss_params["scales"], fb, cfg,
                                      self.gauss_params["normals"] if touch_idx is not None else None, touch_idx,
                                      touch_normals, None, False, _intrinsics_of(self))
    scale_reg = None
    if extras:
        extra, scale_reg = _extra_terms(self, outputs, batch, extras)
        if extra is not None:
            main_loss = main_loss + extra
    return {"main_loss": main_loss, "scale_reg": scale_reg if scale_reg is not None else torch.zeros((), device=rgb.device)}


def get_metrics_dict_fused(self, outputs, batch) -> Dict[str, object]:
    """Drop-in for ``DNSplatterModel.get_metrics_dict`` (dn_model.py:927-1003), which nerfstudio's pipeline calls on
    EVERY training iteration: rgb MSE / PSNR / SSIM11, the seven depth metrics against the sensor depth
    (dn_splatter/metrics.py:109-145), the Gaussian count and ``avg_min_scale`` — PSNR / MSE as device reductions, SSIM
    from the loss kernel's forward (fsgs_ssim_l1_fwd), and ONE device-to-host transfer for all of the floats instead of
    thirteen ``.item()`` calls.  ``rgb_lpips`` needs the reference's pretrained network: the model's own
    ``rgb_metrics.lpips`` module is called when it is there (as the reference does, on every step), NaN otherwise.
    Falls back to the reference's method for down-scaled training images (``num_downscales`` > 0)."""
    from .inference import depth_metrics, psnr, ssim_metric
    rgb = outputs.get("rgb")
    d = self._get_downscale_factor() if hasattr(self, "_get_downscale_factor") else 1
    if not (torch.is_tensor(rgb) and rgb.is_cuda) or d > 1:
        return self._get_metrics_dict_reference(outputs, batch)
    dev = rgb.device
    with torch.no_grad():
        pred = rgb[0] if rgb.dim() == 4 else rgb
        gt = batch["image"].to(dev)
        if gt.dtype == torch.uint8:
            gt = gt.float() / 255.0
        gt = gt[..., :3].to(torch.float32)
        vals = [torch.mean((gt - pred) ** 2), psnr(gt, pred), ssim_metric(gt.contiguous(), pred.contiguous())]
        names = ["rgb_mse", "rgb_psnr", "rgb_ssim"]
        lp = getattr(getattr(self, "rgb_metrics", None), "lpips", None)
        if lp is not None:
            vals.append(lp.to(dev)(pred.permute(2, 0, 1).unsqueeze(0), gt.permute(2, 0, 1).unsqueeze(0)).reshape(()))
            names.append("rgb_lpips")
        if getattr(self.config, "use_depth_loss", False) and "sensor_depth" in batch:
            gd = batch["sensor_depth"].to(dev).to(torch.float32)
            tol = float(getattr(getattr(self, "depth_metrics", None), "tolerance", 0.1))
            vals += list(depth_metrics(outputs["depth"].permute(2, 0, 1), gd.permute(2, 0, 1), tol))
            names += ["depth_abs_rel", "depth_sq_rel", "depth_rmse", "depth_rmse_log", "depth_a1", "depth_a2", "depth_a3"]
        host = torch.stack([v.reshape(()).float() for v in vals]).tolist()  # the step's one transfer
        md = dict(zip(names, host))
        md.setdefault("rgb_lpips", float("nan"))
        md["gaussian_count"] = self.gauss_params["means"].shape[0]
        md["avg_min_scale"] = torch.nanmean(torch.exp(self.gauss_params["scales"][..., -1]))  # (a tensor there too)
    return md


class _NoSlab:
    def rebuild(self, params) -> None:
        pass


class _ParamView:
    """The model's ``gauss_params`` without the inert "normals" entry (rewritten by every get_outputs, dn_model.py:634)."""

    def __init__(self, gauss_params):
        self._p = gauss_params

    def keys(self):
        return [k for k in GAUSS_GROUPS if k in self._p]

    def items(self):
        return [(k, self._p[k]) for k in self.keys()]

    def __getitem__(self, k):
        return self._p[k]

    def __setitem__(self, k, v):
        self._p[k] = v


class _ModelAsTrainer:
    """What splatfacto.DensifyStrategy calls a trainer, over a DNSplatterModel and nerfstudio's ``Optimizers``."""

    def __init__(self, model, optimizers):
        self.model = model
        self.params = _ParamView(model.gauss_params)
        self.optimizers = optimizers.optimizers if hasattr(optimizers, "optimizers") else optimizers
        self.slab = _NoSlab()
        self.device = model.gauss_params["means"].device
        rng = model.__dict__.get("_fsgs_rng")
        if rng is None or rng.device != self.device:
            rng = model.__dict__["_fsgs_rng"] = torch.Generator(device=self.device)
            rng.manual_seed(torch.initial_seed() % (2 ** 63))
        self.rng = rng

    @property
    def step(self) -> int:
        return int(self.model.step)

    def num_gaussians(self) -> int:
        return self.model.gauss_params["means"].shape[0]


def _strategy(model):
    """The model's DensifyStrategy (created on first use from ``model.config``), with the model's statistics and anchor
    mask handed in; ``_sync_back`` returns them."""
    from .splatfacto import DensifyStrategy, SplatfactoConfig
    st = model.__dict__.get("_fsgs_strategy")
    if st is None:
        fields = {f: getattr(model.config, f) for f in SplatfactoConfig.__dataclass_fields__ if hasattr(model.config, f)}
        st = model.__dict__["_fsgs_strategy"] = DensifyStrategy(SplatfactoConfig(**fields),
                                                                num_train_data=int(model.num_train_data))
    for k in ("xys_grad_norm", "vis_counts", "max_2Dsize", "add_mask"):
        setattr(st, k, getattr(model, k, None))
    ls = getattr(model, "last_size", None)
    if ls is not None:
        st.last_size = ls
    return st


def _sync_back(model, st, adapter) -> None:
    for k in ("xys_grad_norm", "vis_counts", "max_2Dsize", "add_mask"):
        setattr(model, k, getattr(st, k))
    n = adapter.num_gaussians()
    nrm = model.gauss_params["normals"] if "normals" in model.gauss_params else None
    if nrm is None or nrm.shape[0] != n:  # the inert group follows N (it is rewritten by the next get_outputs)
        new = torch.nn.Parameter(torch.zeros(n, 3, device=adapter.device))
        model.gauss_params["normals"] = new
        opt = adapter.optimizers.get("normals") if hasattr(adapter.optimizers, "get") else None
        if opt is not None:
            opt.state.clear()
            opt.param_groups[0]["params"] = [new]
    model.__dict__.pop("_fsgs_touch_normals", None)
    model.delete_mask_to_update_knn = None


def after_train_fused(self, step: int):
    """nerfstudio's ``after_train``: the fused node's backward has already accumulated the absgrad
    norm, the visibility counts and the 2-D size maxima into ``self.xys_grad_norm / vis_counts / max_2Dsize``."""
    if getattr(getattr(self, "xys", None), "stats_done", False):
        return None
    return self._after_train_reference(step)


def refinement_after_fused(self, optimizers, step):
    assert step == self.step
    if self.step <= self.config.warmup_length:
        return
    st, ad = _strategy(self), _ModelAsTrainer(self, optimizers)
    st.refinement_after(ad, int(step))
    _sync_back(self, st, ad)


def add_touch_patch_fused(self, optimizers, step):
    assert step == self.step
    if self.step != self.config.add_touch_at or "touch_patches" not in self.kwargs["metadata"]:
        return
    st, ad = _strategy(self), _ModelAsTrainer(self, optimizers)
    md = self.kwargs["metadata"]
    self.added_count = st.add_touch_patch(ad, md["touch_patches"], float(md["gel_scale_factor"]))
    _sync_back(self, st, ad)


def hull_pruning_fused(self, optimizers, step):
    assert step == self.step
    if self.step <= self.config.warmup_length or "visual_hull" not in self.kwargs["metadata"]:
        return
    st, ad = _strategy(self), _ModelAsTrainer(self, optimizers)
    md = self.kwargs["metadata"]
    st.hull_pruning(ad, md["visual_hull"], float(md["scale_factor"]))
    _sync_back(self, st, ad)


def touch_pruning_fused(self, optimizers, step):
    assert step == self.step
    if self.step <= self.config.warmup_length or getattr(self, "add_mask", None) is None:
        return
    st, ad = _strategy(self), _ModelAsTrainer(self, optimizers)
    st.touch_pruning(ad, self.kwargs["metadata"]["touch_patches"])
    _sync_back(self, st, ad)


def patch_all(model_cls) -> None:
    """``patch`` + get_loss_dict, after_train and the four row-moving callbacks on the HIP path (idempotent; every
    original stays reachable as ``_<name>_reference``)."""
    patch(model_cls)
    if getattr(model_cls, "_fsgs_patched_all", False):
        return
    for name, fn in (("get_loss_dict", get_loss_dict_fused), ("get_metrics_dict", get_metrics_dict_fused),
                     ("after_train", after_train_fused),
                     ("refinement_after", refinement_after_fused), ("add_touch_patch", add_touch_patch_fused),
                     ("hull_pruning", hull_pruning_fused), ("touch_pruning", touch_pruning_fused)):
        setattr(model_cls, f"_{name}_reference", getattr(model_cls, name, None))
        setattr(model_cls, name, fn)
    model_cls._fsgs_patched_all = True


# ---------------------------------------------------------------------------------------------------------------------
# nerfstudio's Optimizers: one fsgs_adam_step launch for all Gaussian groups
# ---------------------------------------------------------------------------------------------------------------------
class AdamFuser:
    """Gathers the ``step()`` calls of several single-parameter ``torch.optim.Adam`` objects (nerfstudio builds one per
    parameter group, dn_config.py:36-75) and runs them as ONE libfsgs launch when the last LIVE one has stepped (live =
    its parameter has a gradient; the inert "normals" group never does).  The optimizer objects, their ``state``
    (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, so ``state_dict()`` is unchanged), their param_groups and
    the schedulers bound to them are untouched; only the instance's ``step`` is replaced.  A pending partial set is
    launched by the next ``zero_grad`` of any member, by ``flush()`` and before a ``state_dict()``."""

    def __init__(self, optimizers: Dict[str, torch.optim.Optimizer], names=GAUSS_GROUPS):
        self.members = [optimizers[n] for n in names if n in optimizers]
        self.pending = []
        for opt in self.members:
            assert isinstance(opt, torch.optim.Adam) and len(opt.param_groups) == 1 and \
                len(opt.param_groups[0]["params"]) == 1, "one parameter tensor per optimizer, as nerfstudio builds them"
            g = opt.param_groups[0]
            assert not g.get("amsgrad", False) and g.get("weight_decay", 0) == 0 and not g.get("maximize", False)
            self._wrap(opt)

    def _wrap(self, opt) -> None:
        fuser = self
        orig_zero, orig_sd = opt.zero_grad, opt.state_dict

        def step(closure=None):
            assert closure is None
            opt._step_count = getattr(opt, "_step_count", 0) + 1  # (what torch's LR schedulers look at:
            opt._opt_called = True                                # "optimizer.step() before lr_scheduler.step()")
            p = opt.param_groups[0]["params"][0]
            if p.grad is None:
                return None
            if not p.is_cuda:
                raise RuntimeError("AdamFuser drives libfsgs: parameters must live on the GPU (no CPU fallback)")
            fuser.pending.append(opt)
            if all(o in fuser.pending for o in fuser.live()):
                fuser.flush()
            return None
        step._with_counter = True          # (torch < 2.2) / _wrapped_by_lr_sched (torch >= 2.2): what an LR scheduler
        step._wrapped_by_lr_sched = True   # bound to this optimizer looks for; _step_count is kept above
        opt.step = step

        def zero_grad(*a, **k):
            fuser.flush()
            return orig_zero(*a, **k)

        def state_dict(*a, **k):
            fuser.flush()
            return orig_sd(*a, **k)
        opt.zero_grad, opt.state_dict = zero_grad, state_dict

    def live(self):
        return [o for o in self.members if o.param_groups[0]["params"][0].grad is not None]

    @torch.no_grad()
    def flush(self) -> None:
        if not self.pending:
            return
        from .ops import adam_step_
        todo, self.pending = self.pending, []
        by_key = {}
        for opt in todo:
            g = opt.param_groups[0]
            p = g["params"][0]
            st = opt.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            key = (int(st["step"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
            by_key.setdefault(key, []).append((p, st, float(g["lr"])))
        for (step_no, b1, b2, eps), rows in by_key.items():  # (one launch: the groups step together)
            adam_step_([p.data for p, _, _ in rows], [p.grad for p, _, _ in rows], [s["exp_avg"] for _, s, _ in rows],
                       [s["exp_avg_sq"] for _, s, _ in rows], [lr for _, _, lr in rows], step_no, b1, b2, eps)


def fuse_optimizers(optimizers, names=GAUSS_GROUPS) -> AdamFuser:
    """``optimizers``: nerfstudio's ``Optimizers`` or its ``.optimizers`` dictionary."""
    return AdamFuser(optimizers.optimizers if hasattr(optimizers, "optimizers") else optimizers, names)
