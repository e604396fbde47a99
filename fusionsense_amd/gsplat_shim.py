"""Expose this package under the module names FusionSense imports
(/root/reference/dn_splatter/dn_model.py:29-35), so the reference model file runs unchanged:

    from gsplat.rendering import rasterization
    from gsplat import rasterize_gaussians
    from gsplat.cuda_legacy._torch_impl import quat_to_rotmat
    from gsplat.cuda_legacy._wrapper import num_sh_bases

Call :func:`install` once before importing ``dn_splatter``.
"""
from __future__ import annotations

import sys
import types


def install(force: bool = False) -> None:
    if "gsplat" in sys.modules and not force:
        if getattr(sys.modules["gsplat"], "__fsgs_shim__", False):
            return
        raise RuntimeError("a different `gsplat` is already imported; pass force=True to replace it")
    from . import legacy, rendering

    def mod(name: str, **attrs) -> types.ModuleType:
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    torch_impl = mod("gsplat.cuda_legacy._torch_impl", quat_to_rotmat=legacy.quat_to_rotmat)
    wrapper = mod("gsplat.cuda_legacy._wrapper", num_sh_bases=legacy.num_sh_bases,
                  rasterize_gaussians=legacy.rasterize_gaussians)
    cuda_legacy = mod("gsplat.cuda_legacy", _torch_impl=torch_impl, _wrapper=wrapper)
    rend = mod("gsplat.rendering", rasterization=rendering.rasterization)
    root = mod("gsplat", rendering=rend, cuda_legacy=cuda_legacy, rasterization=rendering.rasterization,
               rasterize_gaussians=legacy.rasterize_gaussians, __version__="1.0.0+fsgs",
               __fsgs_shim__=True)
    root.__path__ = []  # mark as package so `import gsplat.rendering` resolves via sys.modules
    cuda_legacy.__path__ = []
