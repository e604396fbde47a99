"""A minimal training loop around the hot path, for measurement and for multi-GPU data
parallelism.  It mirrors the iteration order of nerfstudio's Trainer
(SURVEY.md A.3): zero_grad -> get_outputs -> losses -> backward -> [grad all-reduce] ->
Adam step -> after_train statistics -> refinement_after every ``refine_every`` steps, with the
8 Adam groups of /root/reference/dn_splatter/dn_config.py:36-75.

Multi-GPU: one process per GPU, every rank holds a full replica of the Gaussians and renders
a different training view (views r, r+W, r+2W, ...).  Gradients of all parameter groups live in
ONE contiguous fp32 slab, so the exchange step is a single RCCL all-reduce over xGMI
(236 B/Gaussian) plus one 12 B/Gaussian all-reduce of the densification statistics; densify/
prune then runs redundantly with a shared RNG seed so N stays identical on every rank.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor

from .scenes import Camera

PARAM_ORDER = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]


@dataclass
class OptimConfig:
    """dn_config.py:36-75 (the 'normals' and 'camera_opt' groups are inert: SURVEY.md A.2/A.3)."""
    lr: Dict[str, float] = field(default_factory=lambda: {
        "means": 1.6e-4, "features_dc": 0.0025, "features_rest": 0.0025 / 20, "opacities": 0.05,
        "scales": 0.005, "quats": 0.001})
    eps: float = 1e-15
    means_lr_final: float = 1.6e-6
    means_lr_max_steps: int = 30000


class GradSlab:
    """All parameter gradients as views into one flat fp32 buffer (one collective per step)."""

    def __init__(self, params: Dict[str, torch.nn.Parameter]):
        self.rebuild(params)

    def rebuild(self, params: Dict[str, torch.nn.Parameter]) -> None:
        total = sum(p.numel() for p in params.values())
        dev = next(iter(params.values())).device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        self.views = {}
        for name in PARAM_ORDER:
            p = params[name]
            v = self.flat[off:off + p.numel()].view_as(p)
            self.views[name] = v
            p.grad = v
            off += p.numel()

    def zero_(self) -> None:
        self.flat.zero_()

    def all_reduce_mean_(self, group=None) -> None:
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "nccl":
                # RCCL averages inside the collective: no extra 2 x 236 B/Gaussian scaling pass
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                self.flat.mul_(1.0 / dist.get_world_size(group))


class SplatTrainer:
    def __init__(self, params: Dict[str, Tensor], device: torch.device, sh_degree: int = 3,
                 optim: Optional[OptimConfig] = None, fused_adam: bool = True, seed: int = 0,
                 strategy=None, fused: bool = True, sh_degree_interval: Optional[int] = None,
                 direct: bool = True):
        self.device = device
        # fused=True: get_outputs as one autograd node (fusionsense_amd/fused.py); False: the
        # reference's op-by-op caller through the drop-in rasterization()/rasterize_gaussians() surface
        self.fused = fused
        # direct=True (fused only): the step's fixed two-node graph (get_outputs, loss) is run forward and
        # backward by hand instead of through torch.autograd (same kernels, same gradients in the slab)
        self.direct = direct
        self.sh_degree = sh_degree
        # sh_degree_to_use = min(step // sh_degree_interval, sh_degree) (dn_model.py:562-565; nerfstudio default
        # interval 1000); None = always the full degree (BASELINE config #2 is quoted at degree 3)
        self.sh_degree_interval = sh_degree_interval
        self.optim_cfg = optim or OptimConfig()
        self.params: Dict[str, torch.nn.Parameter] = {
            k: torch.nn.Parameter(params[k].to(device=device, dtype=torch.float32).contiguous())
            for k in PARAM_ORDER}
        self.fused_adam = fused_adam and device.type == "cuda"
        self.optimizers: Dict[str, torch.optim.Adam] = {}
        for name in PARAM_ORDER:
            self.optimizers[name] = torch.optim.Adam(
                [self.params[name]], lr=self.optim_cfg.lr[name], eps=self.optim_cfg.eps,
                fused=self.fused_adam)
        self.slab = GradSlab(self.params)
        self.step = 0
        self.strategy = strategy  # fusionsense_amd.splatfacto.DensifyStrategy or None
        self.rng = torch.Generator(device=device)
        self.rng.manual_seed(seed)
        self.last_info = None

    # -- schedule: ExponentialDecayScheduler on the means group (dn_config.py:38-41) --------
    def _means_lr(self, step: int) -> float:
        c = self.optim_cfg
        t = min(max(step / c.means_lr_max_steps, 0.0), 1.0)
        return math.exp(math.log(c.lr["means"]) * (1 - t) + math.log(c.means_lr_final) * t)

    def num_gaussians(self) -> int:
        return self.params["means"].shape[0]

    def _sh_degree_now(self, sh_degree_to_use: Optional[int] = None) -> int:
        if sh_degree_to_use is not None:
            return sh_degree_to_use
        if self.sh_degree_interval:
            return min(self.step // self.sh_degree_interval, self.sh_degree)
        return self.sh_degree

    def _frame_state(self, camera: Camera, grad: bool):
        """(after_train statistics target, touch-anchor mask, binary-opacity threshold or None) for this step:
        what the model state supplies to get_outputs in the reference (dn_model.py:492-503, 535-541)."""
        stats, add_mask, bthr = None, None, None
        if self.strategy is not None:
            add_mask = getattr(self.strategy, "add_mask", None)
            cfg = getattr(self.strategy, "cfg", None)
            if cfg is not None:
                from .splatfacto import binary_opacity_active
                if binary_opacity_active(cfg, self.step):
                    bthr = cfg.binary_opacities_threshold
            if grad and self.fused and hasattr(self.strategy, "stats_target"):
                stats = self.strategy.stats_target(self, self.num_gaussians(), self.device, camera)
        return stats, add_mask, bthr

    def forward(self, camera: Camera, sh_degree_to_use: Optional[int] = None, crop_box=None):
        """get_outputs.  In training mode (grad enabled) the strategy supplies what the model state supplies
        in the reference: the binary-opacity schedule (dn_model.py:492-503) and the touch-anchor mask
        (:535-541); ``crop_box`` is honoured only without grad, like ``not self.training`` there (:505-532)."""
        deg = self._sh_degree_now(sh_degree_to_use)
        grad = torch.is_grad_enabled()
        stats, add_mask, bthr = self._frame_state(camera, grad)
        if self.fused:
            from .fused import render_fusionsense_fused
            return render_fusionsense_fused(self.params, camera, sh_degree=deg, device=self.device,
                                            grad_out=self.slab.views if grad else None, stats_out=stats,
                                            add_mask=add_mask, crop_box=crop_box, training=grad,
                                            binary_threshold=bthr)
        from .fusion import render_fusionsense
        return render_fusionsense(self.params, camera, sh_degree=deg, device=self.device, add_mask=add_mask,
                                  crop_box=crop_box, training=grad, binary_threshold=bthr)

    def loss(self, out, target) -> Tensor:
        """Config #2 loss: 0.8*L1 + 0.2*(1-SSIM) on rgb, L1 on depth, L1 on normals."""
        from .losses import rgb_loss
        if self.fused and "depth" in target and out["rgb"].is_cuda:
            from .ops import train_loss
            has_n = "normal" in target
            return train_loss(out["rgb"], target["rgb"], out["depth"], target["depth"],
                              out["normal"] if has_n else None, target["normal"] if has_n else None, 0.2, 0.2, 0.1)
        l = rgb_loss(out["rgb"], target["rgb"])
        if "depth" in target:
            l = l + 0.2 * torch.abs(out["depth"] - target["depth"]).mean()
        if "normal" in target:
            l = l + 0.1 * torch.abs(out["normal"] - target["normal"]).mean()
        return l

    def _fused_adam_step(self) -> None:
        """All six groups in one launch (row N1).  The torch optimizers stay the owners of the state
        (exp_avg / exp_avg_sq per parameter), so densify/prune surgery and checkpoints see the usual
        layout; only their step() is replaced."""
        from .ops import adam_step_
        ps, gs, ms, vs, lrs = [], [], [], [], []
        for name in PARAM_ORDER:
            p = self.params[name]
            opt = self.optimizers[name]
            st = opt.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            st["step"] += 1
            ps.append(p.data); gs.append(self.slab.views[name]); ms.append(st["exp_avg"]); vs.append(st["exp_avg_sq"])
            lrs.append(opt.param_groups[0]["lr"])
        self.adam_steps = getattr(self, "adam_steps", 0) + 1
        adam_step_(ps, gs, ms, vs, lrs, self.adam_steps, 0.9, 0.999, self.optim_cfg.eps)

    def train_step(self, camera: Camera, target: Dict[str, Tensor], optimizer_step: bool = True):
        if getattr(self, "_one", None) is None or self._one.device != self.device:
            self._one = torch.ones((), dtype=torch.float32, device=self.device)
        if self.fused and self.direct and self.device.type == "cuda" and "depth" in target:
            from .fused import fused_step_forward_backward
            stats, add_mask, bthr = self._frame_state(camera, True)
            loss, out = fused_step_forward_backward(self.params, camera, target, self._sh_degree_now(), self.device,
                                                    self.slab.views, self._one, stats_out=stats, add_mask=add_mask,
                                                    binary_threshold=bthr)
        else:
            if not self.fused:
                self.slab.zero_()  # autograd accumulates into the slab views; the fused node overwrites them
            out = self.forward(camera)
            loss = self.loss(out, target)
            loss.backward(gradient=self._one)  # (the default would launch a fill kernel for the seed gradient)
        self.slab.all_reduce_mean_()
        if optimizer_step:
            self.optimizers["means"].param_groups[0]["lr"] = self._means_lr(self.step)
            if self.fused and self.device.type == "cuda":
                self._fused_adam_step()
            else:
                for opt in self.optimizers.values():
                    opt.step()
        if self.strategy is not None:
            self.strategy.after_train(self, out, camera)
            self.strategy.maybe_refine(self)
        self.step += 1
        self.last_info = out["info"]
        return loss.detach(), out
