"""CPU tests of the host side: C-ABI export table, loud failure without a GPU, gsplat shim,
densify schedule/masks against the oracle, reference-helper goldens, data-parallel plumbing."""
import math
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from fusionsense_amd import losses, scenes, touch
from fusionsense_amd.splatfacto import SplatfactoConfig, cull_mask, densify_masks, refine_schedule
from oracle import splatfacto_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_capi_exports_every_declared_symbol():
    """The shared library loads (no GPU needed) and exports every function include/fsgs.h declares;
    the ctypes table covers exactly the same set."""
    from fusionsense_amd import _lib
    header = open(os.path.join(ROOT, "include", "fsgs.h")).read()
    declared = set(re.findall(r"\b(fsgs_[a-z0-9_]+)\s*\(", header))
    declared.discard("fsgs_stream_t")
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in fsgs.h but not exported"
    assert declared == set(_lib.SIGNATURES.keys())
    assert lib.fsgs_version() >= 100
    assert lib.fsgs_error_string(-1).decode().startswith("invalid")
    assert lib.fsgs_sort_scratch_bytes(1 << 20) > (1 << 20) // 2048 * 256 * 4


def test_binding_refuses_a_library_of_another_abi_version(monkeypatch):
    """include/fsgs.h, the built library and the ctypes table carry ONE C-ABI version; load() refuses a library whose
    fsgs_abi_version() differs (a stale libfsgs.so / FSGS_LIB build would otherwise take a stream pointer for a flag)."""
    from fusionsense_amd import _lib
    header = open(os.path.join(ROOT, "include", "fsgs.h")).read()
    declared = int(re.search(r"#define\s+FSGS_ABI_VERSION\s+(\d+)", header).group(1))
    assert declared == _lib.ABI_VERSION == _lib.load().fsgs_abi_version()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", declared + 1)
    with pytest.raises(_lib.FsgsError, match="C-ABI version"):
        _lib.load()
    # the library reads no environment variable (fsgs.h: "no global state")
    import glob
    for src in glob.glob(os.path.join(ROOT, "fusionsense_amd", "csrc", "*.h*")):
        assert "getenv" not in open(src).read(), src


def test_ctypes_signatures_match_the_header():
    """Every ctypes signature has the argument count and the argument KINDS (pointer / 32-bit / 64-bit / float) of its
    declaration in include/fsgs.h — a stray or missing argument would otherwise only show as garbage on the GPU."""
    import ctypes as C
    from fusionsense_amd import _lib
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "fsgs.h")).read(), flags=re.S)
    header = re.sub(r"//[^\n]*", "", header)

    def kind_of_decl(arg):
        a = " ".join(arg.split())
        if "*" in a or "fsgs_stream_t" in a:
            return "ptr"
        if re.match(r"(const )?float\b", a):
            return "f32"
        if re.match(r"(const )?double\b", a):
            return "f64"
        if re.match(r"(const )?(int64_t|size_t|long long|uint64_t)\b", a):
            return "i64"
        if re.match(r"(const )?(int|int32_t|uint32_t|unsigned)\b", a):
            return "i32"
        raise AssertionError(f"unclassified parameter {arg!r}")

    def kind_of_ctype(t):
        if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        if t is C.c_float:
            return "f32"
        if t is C.c_double:
            return "f64"
        if t in (C.c_int64, C.c_uint64, C.c_size_t, C.c_longlong):
            return "i64"
        if t in (C.c_int, C.c_int32, C.c_uint32):
            return "i32"
        raise AssertionError(f"unclassified ctype {t}")

    seen = 0
    for m in re.finditer(r"\b(fsgs_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        if name not in _lib.SIGNATURES:
            continue
        decl = [] if args in ("", "void") else [kind_of_decl(a) for a in args.split(",")]
        bound = [kind_of_ctype(t) for t in _lib.SIGNATURES[name][1]]
        assert decl == bound, f"{name}: header {decl} != ctypes {bound}"
        seen += 1
    assert seen == len(_lib.SIGNATURES)


def test_step_plan_layout_matches_the_c_compiler(tmp_path):
    """fsgs_step_plan (include/fsgs.h: one training step's launches as one argument block) and its ctypes mirror
    _lib.StepPlan agree on the offset of EVERY field and on the size — a shifted field would make the library read a
    stream pointer as a count on the GPU, nowhere else."""
    import ctypes as C
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from fusionsense_amd._lib import AdamGroups, StepPlan
    names = [f[0] for f in StepPlan._fields_]
    body = "".join(f'printf("{n} %zu\\n", offsetof(fsgs_step_plan, {n}));' for n in names)
    body += "".join(f'printf("adam.{f[0]} %zu\\n", offsetof(fsgs_adam_groups, {f[0]}));' for f in AdamGroups._fields_)
    src = tmp_path / "o.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fsgs.h"\nint main(void) {' + body +
                   'printf("sizeof %zu %zu\\n", sizeof(fsgs_step_plan), sizeof(fsgs_adam_groups)); return 0; }\n')
    exe = tmp_path / "o"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in out:
        if not line:
            continue
        k, *v = line.split()
        if k == "sizeof":
            assert int(v[0]) == C.sizeof(StepPlan) and int(v[1]) == C.sizeof(AdamGroups)
        elif k.startswith("adam."):
            assert int(v[0]) == getattr(AdamGroups, k[5:]).offset, k
        else:
            assert int(v[0]) == getattr(StepPlan, k).offset, k
            seen += 1
    assert seen == len(names)
    # ... and the built library was compiled from this header (a stale libfsgs.so would be refused by load() only if its
    # ABI number differs: the block's size catches an edit of the header that forgot the number)
    from fusionsense_amd import _lib
    assert int(_lib.load().fsgs_step_plan_bytes()) == C.sizeof(StepPlan)


def test_header_is_plain_c_and_links(tmp_path):
    """include/fsgs.h is a C header (what a cgo / ctypes / FFI binding of the reference side would include): it compiles
    as C99 with -Wall -Werror, and a C program linked against libfsgs.so calls an entry point."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    from fusionsense_amd import _lib
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include "fsgs.h"\n'
                   'int main(void) { printf("%d %s\\n", fsgs_version(), fsgs_error_string(-1)); return 0; }\n')
    exe = tmp_path / "t"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L", libdir, "-l:libfsgs.so", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) >= 100 and out[1].startswith("invalid")


def test_product_path_fails_loudly_without_gpu():
    """No CPU fallback: CPU tensors are rejected with ValueError by the operator surface."""
    from fusionsense_amd.legacy import rasterize_gaussians
    from fusionsense_amd.rendering import rasterization
    n = 4
    args = dict(means=torch.zeros(n, 3), quats=torch.ones(n, 4), scales=torch.ones(n, 3),
                opacities=torch.ones(n), colors=torch.ones(n, 3), viewmats=torch.eye(4)[None],
                Ks=torch.eye(3)[None], width=16, height=16)
    with pytest.raises(ValueError, match="GPU"):
        rasterization(**args)
    with pytest.raises(ValueError):
        rasterize_gaussians(torch.zeros(n, 2), torch.zeros(n), torch.zeros(n, dtype=torch.int32),
                            torch.zeros(n, 3), torch.zeros(n, dtype=torch.int32), torch.zeros(n, 3),
                            torch.zeros(n, 1), 16, 16, 16)
    with pytest.raises(AssertionError):
        rasterize_gaussians(torch.zeros(n, 2), torch.zeros(n), torch.zeros(n, dtype=torch.int32),
                            torch.zeros(n, 3), torch.zeros(n, dtype=torch.int32), torch.zeros(n, 3),
                            torch.zeros(n, 1), 16, 16, 32)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no module of the product package may import it."""
    pkg = os.path.join(ROOT, "fusionsense_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), fn


def test_gsplat_shim_resolves_reference_imports():
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from fusionsense_amd import gsplat_shim; gsplat_shim.install()\n"
        "from gsplat.rendering import rasterization\n"
        "from gsplat import rasterize_gaussians\n"
        "from gsplat.cuda_legacy._torch_impl import quat_to_rotmat\n"
        "from gsplat.cuda_legacy._wrapper import num_sh_bases\n"
        "import torch\n"
        "assert num_sh_bases(3) == 16 and num_sh_bases(0) == 1 and num_sh_bases(4) == 25\n"
        "R = quat_to_rotmat(torch.tensor([[0.0, 0.0, 0.0, 2.0]]))\n"
        "assert torch.allclose(R[0], torch.diag(torch.tensor([-1.0, -1.0, 1.0])), atol=1e-6)\n"
        "print('ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr


def test_quat_to_rotmat_matches_oracle():
    from fusionsense_amd.legacy import quat_to_rotmat
    from oracle.gsplat_ref import quat_to_rotmat as ref
    q = torch.randn(100, 4)
    assert torch.allclose(quat_to_rotmat(q), ref(q), atol=1e-6)
    Rm = quat_to_rotmat(q)
    assert torch.allclose(Rm @ Rm.transpose(-1, -2), torch.eye(3).expand(100, 3, 3), atol=1e-5)


def test_refine_schedule_matches_reference_conditions():
    cfg = SplatfactoConfig()
    assert not refine_schedule(cfg, 500, 9)["active"]              # warm-up (dn_model.py:328)
    s = refine_schedule(cfg, 600, 9)
    assert s["active"] and s["densify"] and not s["reset_opacity"]  # 600 % 3000 > 9 + 100
    assert not refine_schedule(cfg, 3100, 9)["densify"]            # 3100 % 3000 = 100, not > 109
    assert refine_schedule(cfg, 3100, 9)["reset_opacity"]          # == refine_every
    assert refine_schedule(cfg, 3200, 9)["densify"]
    s = refine_schedule(cfg, 10000, 9)
    assert not s["densify"] and s["cull_only"] and not s["reset_opacity"]


def _state(n, seed):
    g = torch.Generator().manual_seed(seed)
    params = dict(
        means=torch.randn(n, 3, generator=g), scales=torch.log(0.002 + 0.03 * torch.rand(n, 3, generator=g)),
        quats=torch.randn(n, 4, generator=g), features_dc=torch.randn(n, 3, generator=g),
        features_rest=torch.randn(n, 15, 3, generator=g), opacities=2 * torch.randn(n, 1, generator=g))
    state = dict(xys_grad_norm=torch.rand(n, generator=g) * 0.01, vis_counts=1 + torch.randint(0, 5, (n,), generator=g).float(),
                 max_2Dsize=torch.rand(n, generator=g) * 0.2)
    return params, state


@pytest.mark.parametrize("step", [700, 3500, 4500])
def test_densify_and_cull_masks_match_oracle(step):
    cfg = SplatfactoConfig()
    n = 400
    params, state = _state(n, step)
    add_mask = torch.zeros(n, dtype=torch.bool)
    add_mask[:7] = True
    splits, dups = densify_masks(cfg, step, state["xys_grad_norm"], state["vis_counts"], state["max_2Dsize"],
                                 params["scales"], (720, 1280), add_mask)
    adam = {k: dict(exp_avg=torch.zeros_like(v), exp_avg_sq=torch.zeros_like(v)) for k, v in params.items()}
    z = torch.randn(cfg.n_split_samples * int(splits.sum()), 3)
    new_p, new_a, rep = splatfacto_ref.refinement_after(params, adam, dict(state), cfg, step, 9, (720, 1280), z,
                                                        add_mask)
    assert rep["n_split"] == int(splits.sum()) and rep["n_dup"] == int(dups.sum())
    assert rep["n_split"] > 0 and rep["n_dup"] > 0
    # the in-place shrink quirk: some split parents are also duplicated
    if step < 4000:
        assert int((splits & dups).sum()) > 0
    culls = cull_mask(cfg, step, params["opacities"], params["scales"], state["max_2Dsize"], splits)
    n_new = cfg.n_split_samples * rep["n_split"] + rep["n_dup"]
    assert new_p["means"].shape[0] <= n + n_new and int(culls.sum()) <= rep["n_deleted"]


@pytest.mark.parametrize("step", [400, 700, 3100, 3500, 10000])
def test_refinement_host_logic_with_the_row_oracle_matches_reference_execution(step, monkeypatch):
    """DensifyStrategy.refinement_after — schedule, fsgs_refine_rules, the Adam-state bookkeeping around the two launches —
    with the launches replaced by their oracle (oracle/refine_ref.py) against the reference's own refinement_after executed
    in the authoring container (reference_model.npz): pins the oracle the kernels are held to on the GPU
    (tests/test_gpu_densify.py), and the host side of the callback, without a GPU."""
    import numpy as np
    from fusionsense_amd import ops
    from fusionsense_amd.splatfacto import DensifyStrategy
    from fusionsense_amd.trainer import SplatTrainer
    from oracle import refine_ref
    monkeypatch.setattr(ops, "refine_mark", refine_ref.mark)
    monkeypatch.setattr(ops, "refine_move", refine_ref.move)
    d = np.load(os.path.join(ROOT, "tests", "golden", "reference_model.npz"))
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    names = ["means", "scales", "quats", "features_dc", "features_rest", "opacities"]
    st = DensifyStrategy(SplatfactoConfig(), num_train_data=9)
    tr = SplatTrainer({k: t(f"ra.{step}.pre.{k}") for k in names}, torch.device("cpu"), fused=False, strategy=st, seed=0)
    for k in names:
        tr.optimizers[k].state[tr.params[k]] = {"step": torch.tensor(1.0), "exp_avg": t(f"ra.{step}.pre_m.{k}").clone(),
                                                "exp_avg_sq": t(f"ra.{step}.pre_v.{k}").clone()}
    tr.step = step
    st.xys_grad_norm, st.vis_counts, st.max_2Dsize = (t(f"ra.stats.{k}").clone() for k in ("xys_grad_norm", "vis_counts", "max_2Dsize"))
    st.last_size = (720, 1280)
    if step == 3500:
        st.add_mask = torch.zeros(tr.num_gaussians(), dtype=torch.bool)
        st.add_mask[:50] = True
    noise = t(f"ra.{step}.randn")
    real_randn = torch.randn

    def fake_randn(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        if shape == tuple(noise.shape):
            return noise.clone()
        kw.pop("generator", None)
        return real_randn(*size, **kw)

    monkeypatch.setattr(torch, "randn", fake_randn)
    st.refinement_after(tr, step)
    for k in names:
        ref = t(f"ra.{step}.post.{k}")
        assert tr.params[k].shape == ref.shape, (k, tr.params[k].shape, ref.shape, st.last_report)
        assert torch.allclose(tr.params[k].detach(), ref, atol=1e-6), k
        s_ = tr.optimizers[k].state[tr.params[k]]
        assert torch.equal(s_["exp_avg"], t(f"ra.{step}.post_m.{k}")), k
        assert torch.equal(s_["exp_avg_sq"], t(f"ra.{step}.post_v.{k}")), k
    am = d[f"ra.{step}.add_mask_after"]
    if am.size:
        assert np.array_equal(st.add_mask.numpy(), am)
    assert (st.xys_grad_norm is None) == bool(d[f"ra.{step}.stats_cleared"])


def test_refine_rules_layout_matches_the_c_compiler(tmp_path):
    """_lib.RefineRules against ``fsgs_refine_rules`` as gcc lays it out (size and every field's offset)."""
    import ctypes as C
    from fusionsense_amd import _lib
    fields = [f[0] for f in _lib.RefineRules._fields_]
    src = tmp_path / "rules.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fsgs.h"\nint main(void){\n'
                   + "".join(f'printf("{f} %zu\\n", offsetof(fsgs_refine_rules, {f}));\n' for f in fields)
                   + 'printf("sizeof %zu\\n", sizeof(fsgs_refine_rules)); return 0; }\n')
    exe = tmp_path / "rules"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(out["sizeof"]) == C.sizeof(_lib.RefineRules)
    for f in fields:
        assert int(out[f]) == getattr(_lib.RefineRules, f).offset, f


def test_reference_helper_goldens():
    """Product-side mirrors against vectors produced by the reference's own functions
    (tests/golden/make_reference_goldens.py)."""
    g = np.load(os.path.join(GOLD, "reference_helpers.npz"))
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    assert np.array_equal(touch.points_in_non_aabb(t("aabb_pts"), t("aabb_box")).numpy(), g["aabb_mask"])
    assert 0 < g["aabb_mask"].sum() < 500
    assert torch.allclose(touch.rotate_vector_to_vector(t("rvv_v1"), t("rvv_v2")), t("rvv_R"), atol=1e-6)
    assert torch.allclose(touch.matrix_to_quaternion(t("m2q_in")), t("m2q_out"), atol=1e-6)
    assert torch.equal(touch.invert_quaternion(t("invq_in")), t("invq_out"))
    torch.manual_seed(int(g["rq_seed"]))
    assert torch.allclose(scenes.random_quat_tensor(257), t("rq_out"), atol=1e-7)
    assert torch.allclose(scenes.sh_to_rgb(t("sh2rgb_in")), t("sh2rgb_out"))
    assert torch.allclose(scenes.rgb_to_sh(scenes.sh_to_rgb(t("sh2rgb_in"))), t("sh2rgb_in"), atol=1e-5)
    p, gt, rgb, m = t("loss_pred"), t("loss_gt"), t("loss_rgb"), t("loss_mask")
    assert abs(losses.edge_aware_log_l1(p, gt, rgb, None).item() - float(g["edge_logl1_nomask"])) < 1e-6
    assert abs(losses.edge_aware_log_l1(p, gt, rgb, m).item() - float(g["edge_logl1_mask"])) < 1e-6
    assert abs(losses.log_l1(p, gt).item() - float(g["logl1"])) < 1e-6
    assert abs(losses.tv_loss(p).item() - float(g["tv_depth"])) < 1e-6
    assert abs(losses.tv_loss(rgb).item() - float(g["tv_rgb"])) < 1e-6
    assert abs(losses.edge_aware_tv(p, rgb).item() - float(g["edge_tv"])) < 1e-6


def test_touch_gaussian_rows():
    n = 6
    pts = torch.randn(n, 3)
    nrm = torch.nn.functional.normalize(torch.randn(n, 3), dim=-1)
    rows = touch.make_touch_gaussians(pts, nrm, torch.rand(n, 3), 6.34e-5 * 10, 3)
    assert rows["features_rest"].shape == (n, 15, 3) and (rows["opacities"] == 1).all()
    assert torch.allclose(rows["scales"][:, 2], rows["scales"][:, 0] - np.log(3.0), atol=1e-6)
    from fusionsense_amd.legacy import quat_to_rotmat
    z = quat_to_rotmat(rows["quats"]) @ torch.tensor([0.0, 0.0, 1.0])
    assert torch.allclose(z, nrm, atol=1e-5), "local +z is rotated onto the contact normal"


def test_ssim_properties():
    a = torch.rand(3, 40, 40)
    assert abs(losses.ssim(a, a).item() - 1.0) < 1e-6
    assert losses.ssim(a, torch.rand(3, 40, 40)).item() < 0.2
    assert abs(losses.rgb_loss(a.permute(1, 2, 0), a.permute(1, 2, 0)).item()) < 1e-6


def test_view_sharding_is_a_partition():
    """rank r takes views r, r+W, ...: every view exactly once per W-step window."""
    world, n_views = 4, 8
    seen = [(s * world + r) % n_views for s in range(2) for r in range(world)]
    assert sorted(seen) == list(range(n_views))


_DP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from fusionsense_amd.trainer import GradSlab, PARAM_ORDER
from fusionsense_amd.splatfacto import DensifyStrategy
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=rank, world_size=world)
torch.manual_seed(0)
n = 50
shapes = dict(means=(n,3), scales=(n,3), quats=(n,4), features_dc=(n,3), features_rest=(n,15,3), opacities=(n,1))
params = {k: torch.nn.Parameter(torch.randn(*shapes[k])) for k in PARAM_ORDER}
slab = GradSlab(params)
assert slab.flat.numel() == n * 59, "236 B/Gaussian gradient slab"
for i, k in enumerate(PARAM_ORDER):
    params[k].grad.copy_(torch.full(shapes[k], float((rank + 1) * (i + 1))))
slab.all_reduce_mean_()
for i, k in enumerate(PARAM_ORDER):
    want = (i + 1) * (1 + world) / 2.0
    assert torch.allclose(params[k].grad, torch.full(shapes[k], want)), (k, params[k].grad.flatten()[0], want)
# autograd accumulates IN PLACE into the slab views
slab.zero_()
loss = sum((p * p).sum() for p in params.values())
loss.backward()
assert params["means"].grad.data_ptr() == slab.views["means"].data_ptr()
assert torch.allclose(slab.views["quats"], 2 * params["quats"].data)
# densification statistics: SUM / SUM of (count-1) / MAX
st = DensifyStrategy()
st.xys_grad_norm = torch.full((n,), float(rank + 1)); st.vis_counts = torch.full((n,), 1.0 + rank + 1)
st.max_2Dsize = torch.full((n,), 0.1 * (rank + 1))
st._all_reduce_stats()
assert torch.allclose(st.xys_grad_norm, torch.full((n,), world * (world + 1) / 2.0))
assert torch.allclose(st.vis_counts, torch.full((n,), 1.0 + world * (world + 1) / 2.0))
assert torch.allclose(st.max_2Dsize, torch.full((n,), 0.1 * world))
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_data_parallel_allreduce_gloo_world2():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = _DP_WORKER % (ROOT, port)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0 and "ok" in out, err[-2000:]


_COMM_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from fusionsense_amd import comm
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=rank, world_size=world)
assert comm.world() == world and comm.rank() == rank and comm.exchange()
# all_reduce(AVG), in stream order and asynchronous
t = torch.arange(6, dtype=torch.float32) * (rank + 1)
assert comm.all_reduce_mean_(t) is None and torch.equal(t, torch.arange(6, dtype=torch.float32) * 1.5)
t = torch.full((5,), float(rank)); fin = comm.all_reduce_mean_(t, async_op=True); fin(); assert torch.equal(t, torch.full((5,), 0.5))
assert comm.all_reduce_mean_(torch.empty(0)) is None
# reduce_scatter(AVG): the own shard of the mean
full = torch.arange(8, dtype=torch.float32) + 10.0 * rank
out = torch.empty(4); comm.reduce_scatter_mean(out, full)
assert torch.equal(out, torch.arange(8, dtype=torch.float32)[4 * rank:4 * rank + 4] + 5.0)
# in-place all-gather of the ranks' own slices
flat = torch.zeros(8); flat[4 * rank:4 * rank + 4] = rank + 1.0
comm.all_gather_shards_(flat, 4 * rank, 4 * rank + 4)
assert torch.equal(flat, torch.tensor([1., 1, 1, 1, 2, 2, 2, 2]))
# asynchronous gather of per-rank blocks, rank order
own = torch.full((3, 4), float(rank + 1)); gathered = torch.empty(world, 3, 4)
comm.all_gather_blocks_async(gathered, own).wait()
assert torch.equal(gathered[0], torch.ones(3, 4)) and torch.equal(gathered[1], 2 * torch.ones(3, 4))
m = torch.tensor([float(rank), 5.0 - rank]); comm.all_reduce_(m, "max"); assert torch.equal(m, torch.tensor([1.0, 5.0]))
s_ = torch.tensor([1.0 + rank]); comm.all_reduce_(s_, "sum"); assert float(s_) == 3.0
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_collectives_single_code_path_gloo_world2():
    """fusionsense_amd/comm.py: the collectives of the data-parallel step are ONE code path — the calls an RCCL job
    makes (all_reduce AVG, reduce_scatter_tensor AVG, in-place all_gather_into_tensor, the asynchronous block gather)
    are the calls the gloo tests make; no module branches on the backend any more."""
    import socket
    for mod in ("trainer.py", "comm.py", "splatfacto.py"):
        src = open(os.path.join(ROOT, "fusionsense_amd", mod)).read()
        assert "get_backend" not in src, f"{mod} must not choose its collectives by backend"
    for mod in ("trainer.py", "splatfacto.py"):
        src = open(os.path.join(ROOT, "fusionsense_amd", mod)).read()
        assert not re.search(r"\bdist\.(all_|reduce_|broadcast)", src), f"{mod} must issue collectives through comm.py"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = _COMM_WORKER % (ROOT, port)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0 and "ok" in out, err[-2000:]


_DEFER_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from fusionsense_amd.trainer import SplatTrainer, PARAM_ORDER, GEOMETRY_GROUPS, FEATURE_GROUPS
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=rank, world_size=world)
torch.manual_seed(0)
n = 40
shapes = dict(means=(n,3), scales=(n,3), quats=(n,4), features_dc=(n,3), features_rest=(n,15,3), opacities=(n,1))
init = {k: torch.randn(*shapes[k]) for k in PARAM_ORDER}
dev = torch.device("cpu")
a = SplatTrainer(init, dev, fused=False)   # feature half of the exchange + its Adam deferred to the next frame
b = SplatTrainer(init, dev, fused=False)   # everything at once
b.defer_features = False
assert a.slab.split == n * 11 and a.slab.flat.numel() == n * 59
def grads(tr, it):
    for i, k in enumerate(PARAM_ORDER):
        tr.slab.views[k].copy_(torch.sin(tr.params[k].data * (i + 1) + it) * (rank + 1) + 0.1 * rank)
for it in range(4):
    grads(a, it); grads(b, it)
    feat_before = {k: a.params[k].data.clone() for k in FEATURE_GROUPS}
    geom_before = {k: a.params[k].data.clone() for k in GEOMETRY_GROUPS}
    a._reduce_and_step(True)
    assert a._pending is not None
    for k in FEATURE_GROUPS:
        assert torch.equal(a._params[k].data, feat_before[k]), "features must wait for the flush"
    for k in GEOMETRY_GROUPS:
        assert not torch.equal(a._params[k].data, geom_before[k]), "geometry is stepped at once"
    a.flush(); a.flush()
    assert a._pending is None
    b._reduce_and_step(True)
    a.step += 1; b.step += 1
    for k in PARAM_ORDER:
        assert torch.equal(a.params[k].data, b.params[k].data), (it, k)
        assert torch.equal(a.slab.views[k], b.slab.views[k]), (it, k)
        sa, sb = a.optimizers[k].state[a.params[k]], b.optimizers[k].state[b.params[k]]
        assert torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and float(sa["step"]) == float(sb["step"]) == it + 1
# without an optimizer step nothing is deferred
grads(a, 9); a._reduce_and_step(False); assert a._pending is None
# every rank holds the same parameters
for k in PARAM_ORDER:
    t = a.params[k].data.clone(); dist.broadcast(t, 0)
    assert torch.equal(t, a.params[k].data), k
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_deferred_feature_exchange_equals_plain_step_gloo_world2():
    """Data-parallel step with the SH features' share of the all-reduce (and their Adam) deferred to the next
    frame (trainer._reduce_and_step / flush) against the plain all-reduce + step: identical parameters, gradients
    and optimizer state on both ranks over several steps."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = _DEFER_WORKER % (ROOT, port)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0 and "ok" in out, err[-2000:]


def test_binary_opacity_schedule_matches_reference_transcription():
    """dn_model.py:492-499: after the warm-up, every step except the reset steps and the 200 after them."""
    from fusionsense_amd.splatfacto import SplatfactoConfig, binary_opacity_active, binary_opacity_write_
    from oracle.fusion_ref import binary_opacity_step
    cfg = SplatfactoConfig()
    for step in range(0, 9500):
        assert binary_opacity_active(cfg, step) == binary_opacity_step(step), step
    assert not binary_opacity_active(cfg, 500) and binary_opacity_active(cfg, 501)
    assert not binary_opacity_active(cfg, 3000) and not binary_opacity_active(cfg, 3200) and binary_opacity_active(cfg, 3201)
    off = SplatfactoConfig(use_binary_opacities=False)
    assert not any(binary_opacity_active(off, s) for s in range(0, 4000, 7))
    o = torch.nn.Parameter(torch.tensor([[0.89999], [0.9], [5.0], [-3.0]]))
    binary_opacity_write_(o, cfg.binary_opacities_threshold)
    assert o.detach().flatten().tolist() == [0.0, 1.0, 1.0, 0.0] and o.requires_grad


def test_oriented_box_and_empty_outputs():
    """OrientedBox.within against the homogeneous-transform statement of nerfstudio's method; empty outputs."""
    from fusionsense_amd.crop import OrientedBox, get_empty_outputs, crop_params
    from fusionsense_amd import scenes
    from oracle.fusion_ref import obb_within
    g = torch.Generator().manual_seed(3)
    for _ in range(5):
        R = scenes.look_at_c2w(torch.randn(3, generator=g), torch.randn(3, generator=g))[:3, :3]
        T = torch.randn(3, generator=g) * 0.3
        S = torch.rand(3, generator=g) + 0.2
        pts = torch.randn(4000, 3, generator=g)
        box = OrientedBox(R, T, S)
        a, b = box.within(pts), obb_within(R, T, S, pts)
        assert a.dtype == torch.bool and 0 < int(a.sum()) < 4000
        assert int((a != b).sum()) == 0
    out = get_empty_outputs(7, 5, torch.tensor([0.1, 0.2, 0.3]))
    assert out["rgb"].shape == (5, 7, 3) and torch.equal(out["rgb"][4, 6], torch.tensor([0.1, 0.2, 0.3]))
    assert out["depth"].shape == (5, 7, 1) and float(out["depth"].min()) == 10.0
    assert out["accumulation"].shape == (5, 7, 1) and float(out["accumulation"].abs().max()) == 0.0
    p = {"means": torch.arange(12.0).view(4, 3), "opacities": torch.arange(4.0).view(4, 1)}
    c = crop_params(p, torch.tensor([True, False, True, False]))
    assert c["means"].shape == (2, 3) and c["opacities"].flatten().tolist() == [0.0, 2.0]


def test_oracle_touch_anchor_detach_and_crop():
    """The oracle's get_outputs with add_mask (anchors get no means / opacity / scale gradient, but do get
    quaternion and colour gradients) and with an eval crop (= rendering the kept subset; nothing kept ->
    empty outputs)."""
    from fusionsense_amd import scenes
    from oracle.fusion_ref import render_fusionsense as render_ref
    params, cam = scenes.cube_scene(60, seed=2)
    cam = scenes.Camera(cam.c2w, 32.0, 32.0, 16.0, 16.0, 32, 32)
    add_mask = torch.zeros(60, dtype=torch.bool)
    add_mask[::3] = True
    p = {k: v.clone().double().requires_grad_(True) for k, v in params.items()}
    out = render_ref(p, cam, sh_degree=3, add_mask=add_mask)
    (out["rgb"].sum() + out["depth"].sum() + out["normal"].sum()).backward()
    for k in ("means", "opacities", "scales"):
        assert float(p[k].grad[add_mask].abs().max()) == 0.0, k
        assert float(p[k].grad[~add_mask].abs().max()) > 0.0, k
    assert float(p["quats"].grad[add_mask].abs().max()) > 0.0
    assert float(p["features_dc"].grad[add_mask].abs().max()) > 0.0
    box = (torch.eye(3), torch.zeros(3), torch.tensor([0.6, 2.0, 2.0]))
    with torch.no_grad():
        pc = {k: v.clone().double() for k, v in params.items()}
        keep = (pc["means"][:, 0].abs() < 0.3)
        full = render_ref(pc, cam, sh_degree=3, crop_box=box, training=False)
        sub = render_ref({k: v[keep] for k, v in pc.items()}, cam, sh_degree=3)
        assert 0 < int(keep.sum()) < 60
        assert torch.equal(full["rgb"], sub["rgb"]) and torch.equal(full["depth"], sub["depth"])
        same = render_ref(pc, cam, sh_degree=3, crop_box=box, training=True)      # ignored while training
        ref = render_ref(pc, cam, sh_degree=3)
        assert torch.equal(same["rgb"], ref["rgb"])
        none = render_ref(pc, cam, sh_degree=3, crop_box=(torch.eye(3), torch.full((3,), 50.0), torch.ones(3)), training=False)
        assert set(none) == {"rgb", "depth", "accumulation", "background"} and float(none["depth"].min()) == 10.0


def test_on_disk_formats_roundtrip(tmp_path):
    """Row N3: PLY / PCD parsing (ascii and binary), seed initialisation of populate_modules (dn_model.py:196-300),
    the ns-export gaussian-splat PLY layout, and the nerfstudio checkpoint dictionary."""
    import numpy as np
    from fusionsense_amd import io as fio, scenes
    from fusionsense_amd.scenes import C0
    g = torch.Generator().manual_seed(0)
    n = 257
    pts = torch.randn(n, 3, generator=g)
    col = torch.randint(0, 256, (n, 3), generator=g).to(torch.uint8)
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    # --- PLY: binary written by us, ascii written by hand; both parse to the same arrays
    props = {"x": pts[:, 0].numpy(), "y": pts[:, 1].numpy(), "z": pts[:, 2].numpy(),
             "nx": nrm[:, 0].numpy(), "ny": nrm[:, 1].numpy(), "nz": nrm[:, 2].numpy(),
             "red": col[:, 0].numpy(), "green": col[:, 1].numpy(), "blue": col[:, 2].numpy()}
    fio.write_ply(str(tmp_path / "seed.ply"), props)
    with open(tmp_path / "seed_ascii.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment hand written\nelement vertex %d\n" % n)
        for k in ("x", "y", "z", "nx", "ny", "nz"):
            f.write(f"property float {k}\n")
        for k in ("red", "green", "blue"):
            f.write(f"property uchar {k}\n")
        f.write("element face 0\nproperty list uchar int vertex_indices\nend_header\n")
        for i in range(n):
            f.write(" ".join(repr(float(props[k][i])) for k in ("x", "y", "z", "nx", "ny", "nz")) + " " +
                    " ".join(str(int(props[k][i])) for k in ("red", "green", "blue")) + "\n")
    a, b = fio.read_ply(str(tmp_path / "seed.ply")), fio.read_ply(str(tmp_path / "seed_ascii.ply"))
    assert list(a) == list(props) == list(b)
    for k in props:
        assert a[k].dtype == props[k].dtype and np.array_equal(a[k], props[k]) and np.array_equal(b[k], props[k]), k
    # --- PCD ascii
    with open(tmp_path / "patch.pcd", "w") as f:
        f.write("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 5\nHEIGHT 1\n"
                "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 5\nDATA ascii\n")
        for i in range(5):
            f.write(f"{pts[i, 0].item()!r} {pts[i, 1].item()!r} {pts[i, 2].item()!r}\n")
    pc = fio.read_pcd(str(tmp_path / "patch.pcd"))
    assert np.array_equal(np.stack([pc["x"], pc["y"], pc["z"]], -1), pts[:5].numpy())
    # --- seeds -> Gaussians
    P, C, Nn = fio.load_seed_points(str(tmp_path / "seed.ply"))
    assert torch.equal(P, pts) and torch.equal(C, col.float()) and torch.allclose(Nn, nrm)
    d = torch.cdist(pts.double(), pts.double())
    d.fill_diagonal_(float("inf"))
    want = d.topk(3, largest=False).values.mean(-1, keepdim=True).float()
    assert torch.allclose(fio.knn_mean_distance(pts, 3), want, rtol=1e-4, atol=1e-6)
    gp = fio.init_gauss_params(P, C, None, sh_degree=3, generator=torch.Generator().manual_seed(1))
    assert gp["features_rest"].shape == (n, 15, 3) and float(gp["features_rest"].abs().max()) == 0.0
    assert torch.allclose(gp["features_dc"], (C / 255 - 0.5) / C0)
    assert torch.allclose(gp["scales"], torch.log(want).repeat(1, 3), rtol=1e-4, atol=1e-5)
    assert torch.allclose(gp["opacities"], torch.full((n, 1), -2.1972246), atol=1e-5)
    assert torch.allclose(gp["quats"].norm(dim=-1), torch.ones(n), atol=1e-5)
    gn = fio.init_gauss_params(P, C, Nn, sh_degree=3)
    assert torch.allclose(gn["scales"][:, 2], torch.log(want[:, 0] / 10), rtol=1e-4, atol=1e-5)
    from fusionsense_amd.legacy import quat_to_rotmat
    z_axis = quat_to_rotmat(gn["quats"] / gn["quats"].norm(dim=-1, keepdim=True))[:, :, 2]
    assert (z_axis - Nn).abs().max().item() < 1e-4, "the thin axis is turned onto the seed normal"
    # --- splat PLY: header order, channel-major f_rest, non-finite rows dropped, exact round trip
    gp["features_rest"] = torch.randn(n, 15, 3, generator=g)
    gp["means"][3, 1] = float("nan")
    kept = fio.export_gaussian_splat_ply(gp, str(tmp_path / "splat.ply"))
    assert kept == n - 1
    raw = fio.read_ply(str(tmp_path / "splat.ply"))
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert list(raw) == names and all(v.dtype == np.float32 for v in raw.values())
    ok = torch.ones(n, dtype=torch.bool)
    ok[3] = False
    assert np.array_equal(raw["f_rest_1"], gp["features_rest"][ok][:, 1, 0].numpy())       # channel 0, band 2
    assert np.array_equal(raw["f_rest_15"], gp["features_rest"][ok][:, 0, 1].numpy())      # channel 1, band 1
    back = fio.import_gaussian_splat_ply(str(tmp_path / "splat.ply"))
    for k in gp:
        assert torch.equal(back[k], gp[k][ok]), k


def test_checkpoint_roundtrip_resizes_like_splatfacto(tmp_path):
    from fusionsense_amd import io as fio, scenes
    from fusionsense_amd.trainer import SplatTrainer
    dev = torch.device("cpu")
    pa, _ = scenes.cube_scene(50, seed=1)
    pb, _ = scenes.cube_scene(31, seed=2)
    a = SplatTrainer(pa, dev, fused=False)
    for k, p in a.params.items():           # one hand-made optimizer step so that the Adam moments exist
        p.grad = torch.full_like(p, 0.01)
    for opt in a.optimizers.values():
        opt.step()
    a.step = 1234
    fio.save_checkpoint(a, str(tmp_path / "step-000001234.ckpt"))
    ck = torch.load(str(tmp_path / "step-000001234.ckpt"), weights_only=False)
    assert set(ck) >= {"step", "pipeline", "optimizers"} and "_model.gauss_params.means" in ck["pipeline"]
    b = SplatTrainer(pb, dev, fused=False)
    assert fio.load_checkpoint(b, str(tmp_path / "step-000001234.ckpt")) == 1234
    assert b.num_gaussians() == 50 and b.slab.flat.numel() == 50 * 59
    for k in a.params:
        assert torch.equal(a.params[k].detach(), b.params[k].detach()), k
        sa, sb = a.optimizers[k].state[a.params[k]], b.optimizers[k].state[b.params[k]]
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert b.params[k].grad is b.slab.views[k]
    # nerfstudio's conventions: "step" is the COMPLETED step (resume at step + 1), gauss_params.normals is a key,
    # and a file without this package's private keys still restores the Adam bias-correction count
    assert ck["step"] == 1233 and ck["pipeline"]["_model.gauss_params.normals"].shape == (50, 3)
    ns = {k: v for k, v in ck.items() if k not in ("fsgs_next_step", "adam_steps")}
    torch.save(ns, str(tmp_path / "ns.ckpt"))
    c = SplatTrainer(pb, dev, fused=False)
    assert fio.load_checkpoint(c, str(tmp_path / "ns.ckpt")) == 1234 and c.adam_steps == 1


def test_unfused_train_step_flushes_deferred_features_before_clearing_the_slab(monkeypatch):
    """with the deferred feature update (data-parallel step order, forced on one rank) the op-by-op
    trainer zeroes the gradient slab at the top of train_step — the previous step's pending feature update reads
    that slab, so it must have landed first.  Driven through train_step itself with a CPU stand-in for get_outputs."""
    from fusionsense_amd.trainer import PARAM_ORDER, SplatTrainer

    class T(SplatTrainer):
        def forward(self, camera, **kw):
            self.flush()
            val = sum(((p * (i + 1)).sin() * (1.0 + 0.1 * self.step)).sum() for i, p in enumerate(self.params.values()))
            return {"val": val, "info": None}

        def loss(self, out, target, camera=None):
            return out["val"]

    torch.manual_seed(0)
    n = 30
    shapes = dict(means=(n, 3), scales=(n, 3), quats=(n, 4), features_dc=(n, 3), features_rest=(n, 15, 3), opacities=(n, 1))
    init = {k: torch.randn(*shapes[k]) for k in PARAM_ORDER}
    init0 = {k: v.clone() for k, v in init.items()}
    cpu = torch.device("cpu")
    a, b = T(init, cpu, fused=False), T(init, cpu, fused=False)
    a.force_split_step = True
    for it in range(4):
        a.train_step(None, {})
        b.train_step(None, {})
        assert a._pending is not None and b._pending is None
    a.flush()
    for k in PARAM_ORDER:
        assert torch.equal(a.params[k].data, b.params[k].data), k
    for k in PARAM_ORDER:  # the trainer owns a copy: the caller's tensors are untouched
        assert torch.equal(init[k], init0[k]) and not torch.equal(init[k], a.params[k].data), k


def test_dataparser_scene_directory(tmp_path):
    """transforms.json -> cameras (frames in file-name order, y/z axis flip, no re-orientation, max |t| scaled to 1),
    seed + hull points and touch patches moved by the same transform and scale, per-view batches in the units
    get_loss_dict expects (normal_nerfstudio.py:136-725, dn_dataset.py:110-243) — against an independent numpy
    computation on a generated scene directory."""
    from fusionsense_amd import dataparser as dp, touch
    rng = np.random.default_rng(0)
    from helpers import write_scene_dir
    names, c2ws, pts, col, hull, raws = write_scene_dir(tmp_path, rng)
    out = dp.parse_transforms(tmp_path, load_touches=True)
    order = sorted(names)
    assert [p.stem for p in out.image_filenames] == order and [p.stem for p in out.mask_filenames] == order
    assert [Path_.stem for Path_ in out.metadata["normal_filenames"]] == order
    tmax = max(np.abs(c2ws[n][:3, 3]).max() for n in names)
    assert abs(out.dataparser_scale - 1.0 / tmax) < 1e-6
    assert torch.equal(out.dataparser_transform, torch.eye(4)[:3])
    for cam, n in zip(out.cameras, order):
        T = c2ws[n].astype(np.float32)
        exp = T[:3, :4].copy()
        exp[:, 1:3] *= -1
        exp[:, 3] *= out.dataparser_scale
        assert np.allclose(cam.c2w.numpy(), exp, atol=1e-6)
        assert (cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height) == (20.0, 21.0, 8.0, 6.0, 16, 12)
    assert max(float(c.c2w[:, 3].abs().max()) for c in out.cameras) == pytest.approx(1.0, abs=1e-6)
    md = out.metadata
    assert np.allclose(md["points3D_xyz"].numpy(), pts * out.dataparser_scale, atol=1e-6)
    assert np.array_equal(md["points3D_rgb"].numpy(), col)
    assert np.allclose(md["visual_hull"].numpy(), hull * out.dataparser_scale, atol=1e-6)
    n3 = md["points3D_normals"]
    assert n3.shape == (60, 3) and torch.allclose(n3.norm(dim=-1), torch.ones(60), atol=1e-5)
    # touch patches
    assert len(md["touch_patches"]) == 2 and md["gel_scale_factor"] == 6.34e-5
    for patch, (raw, mask, nrm, T) in zip(md["touch_patches"], raws):
        r = raw[::5].astype(np.float64)
        c = r.copy()
        c[:, :2] -= r.mean(0)[:2]
        c *= 6.34e-5
        c = (c @ T[:3, :3].T + T[:3, 3]) * out.dataparser_scale
        m = mask[::5]
        assert np.allclose(patch["points_xyz"].numpy(), c[m], atol=1e-6)
        assert np.allclose(patch["normals"].numpy(), nrm[::5][m] @ T[:3, :3].T + T[:3, 3], atol=1e-5)
        assert patch["bbox"].shape == (8, 3) and patch["points_rgb"].abs().sum() == 0
        inside = touch.points_in_non_aabb(patch["points_xyz"], patch["bbox"])
        assert float(inside.float().mean()) > 0.9, "the oriented box is built around the patch"
    # a plane of seed points: the PCA normal is the plane's normal (either sign)
    g = torch.Generator().manual_seed(0)
    plane = torch.cat([torch.rand(200, 2, generator=g) * 0.2, torch.zeros(200, 1)], -1)
    assert bool((dp.estimate_normals(plane)[:, 2].abs() > 0.999).all())
    # per-view batch
    b = dp.load_batch(out, 1)
    assert b["image"].shape == (12, 16, 3) and b["image"].dtype == torch.float32 and 0 <= float(b["image"].min()) <= float(b["image"].max()) <= 1
    assert b["mask"].shape == (12, 16, 1) and b["mask"].dtype == torch.bool
    from PIL import Image
    raw_d = np.array(Image.open(tmp_path / "depth" / f"{order[1]}.png")).astype(np.float64)
    assert torch.allclose(b["sensor_depth"][..., 0].double(), torch.from_numpy(raw_d * 1e-3 * out.dataparser_scale), atol=1e-6)
    assert b["normal"].shape == (12, 16, 3) and 0 <= float(b["normal"].min()) and float(b["normal"].max()) <= 1
    gp = dp.seed_gauss_params(out)
    assert gp["means"].shape == (60, 3) and gp["features_rest"].shape == (60, 15, 3) and gp["quats"].shape == (60, 4)
    assert torch.allclose(gp["scales"][:, 2], gp["scales"][:, 0] + math.log(0.1), atol=1e-5), "seed normals: 10x thinner z axis"


def test_inference_helpers_match_reference_goldens():
    """fusionsense_amd/inference.py against outputs of the reference's own dn_splatter/metrics.py,
    utils/camera_utils.py and export_mesh.find_depth_edges (tests/golden/make_reference_inference_goldens.py)."""
    from fusionsense_amd import inference as inf
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_inference.npz"))
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    got = torch.stack(list(inf.depth_metrics(t("dm_pred"), t("dm_gt")))).double()
    assert torch.allclose(got, t("dm_out"), rtol=1e-6, atol=1e-7)
    got = torch.stack(list(inf.normal_metrics(t("nm_pred"), t("nm_gt")))).double()
    assert torch.allclose(got, t("nm_out"), rtol=1e-6, atol=1e-7)
    assert torch.allclose(inf.mean_angular_error(t("nm_pred"), t("nm_gt")), t("mae_map"), atol=1e-6)
    fx, fy, cx, cy = [float(v) for v in d["bp_intr"]]
    H, W = d["bp_depth"].shape[:2]
    assert torch.equal(inf.get_camera_coords((W, H)), t("bp_coords"))
    pts, _ = inf.get_means3d_backproj(t("bp_depth"), fx, fy, cx, cy, (W, H), t("bp_c2w"))
    assert torch.allclose(pts, t("bp_points"), atol=1e-5)
    p2, c2 = inf.get_colored_points_from_depth(t("bp_depth"), t("bp_rgb"), t("bp_c2w"), fx, fy, cx, cy, (W, H), t("bp_idx"))
    assert torch.allclose(p2, t("bp_points_sel"), atol=1e-5) and torch.equal(c2, t("bp_colors_sel"))
    assert torch.allclose(inf.project_pix(pts, fx, fy, cx, cy, t("bp_c2w"), return_z_depths=True), t("bp_uvz"), atol=1e-4)
    assert torch.equal(inf.find_depth_edges(t("edge_depth"), 0.01, 3), t("edge_out"))
    assert torch.equal(inf.find_depth_edges(t("edge_depth"), 0.05, 1), t("edge_out_1"))
    assert float(t("edge_out").sum()) > 0 and float(t("edge_out_1").sum()) < float(t("edge_out").sum())


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """bench.py never measures a different job than the one asked for: a launcher-provided WORLD_SIZE that differs
    from --gpus is refused (exit 2) before anything touches the GPU, and --gpus N without a launcher over RCCL on a
    box with fewer than N GPUs fails loudly instead of silently running one rank."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE 1 != --gpus 8" in r.stderr and not r.stdout.strip()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FSGS_DIST_BACKEND")}
    if torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 2 and "needs 8 visible GPUs" in r.stderr and not r.stdout.strip()


def test_lazy_meta_copies_never_lose_the_lazy_keys():
    """rendering.LazyMeta: every way of handing the dictionary's content on (copy, copy.copy, deepcopy, pickling,
    dict(meta), {**meta}, update) first builds the lazy list entries — a copy can never lack a key ``in`` reported."""
    import copy
    import pickle
    from fusionsense_amd.rendering import LazyMeta
    built = []

    def mk():
        def build():
            built.append(1)
            return {"isect_ids": [1], "flatten_ids": [2], "isect_offsets": [3], "legacy_rule_diff": 0}
        return LazyMeta({"radii": 7}, build)

    m = mk()
    assert "isect_offsets" in m and not built and m["radii"] == 7 and not built  # eager reads build nothing
    for clone in (lambda x: x.copy(), copy.copy, copy.deepcopy, lambda x: pickle.loads(pickle.dumps(x)), dict,
                  lambda x: {**x}, lambda x: (lambda d: (d.update(x), d)[1])({})):
        m = mk()
        c = clone(m)
        assert type(c) is dict and set(c) == {"radii", "isect_ids", "flatten_ids", "isect_offsets", "legacy_rule_diff"}
    assert len(built) == 7


@pytest.mark.parametrize("variant", ["plain", "ragged"])
def test_dataparser_matches_reference_execution(tmp_path, variant, monkeypatch):
    """fusionsense_amd/dataparser.py against what the REFERENCE's own NormalNerfstudio._generate_dataparser_outputs
    (normal_nerfstudio.py:136-725) and GDataset.get_metadata (dn_dataset.py:110-243) produced on the same scene
    directory (tests/golden/make_reference_dataparser_goldens.py; the directory is rewritten here from the same
    seed): frame order (file-name sort, natural sort, the second application of the permutation), camera poses after
    the axis flip and the auto scale, per-frame intrinsics, seed / hull points, what happens to the seed normals,
    touch patches (.npy and .pcd masks, 2- and 3-channel normals, oriented boxes), applied_transform / applied_scale,
    and the per-view sensor depth and normal images."""
    from pathlib import Path
    from fusionsense_amd import dataparser as dp
    from helpers import write_scene_dir
    sys.path.insert(0, GOLD)
    try:
        from make_reference_dataparser_goldens import fake_pca_normals
    finally:
        sys.path.remove(GOLD)
    d = np.load(os.path.join(GOLD, "reference_dataparser.npz"))
    g = lambda k: d[f"{variant}.{k}"]  # noqa: E731
    write_scene_dir(tmp_path, np.random.default_rng(int(g("seed"))), variant=variant)
    # open3d's PCA normals cannot be executed by the generator either: both sides use the same fixed function, the
    # golden pins what the reference does with them
    monkeypatch.setattr(dp, "estimate_normals",
                        lambda pts, **k: torch.from_numpy(fake_pca_normals(pts.numpy().astype(np.float64)).astype(np.float32)))
    fmt = "opengl" if bool(g("normal_format_is_opengl")) else "opencv"
    out = dp.parse_transforms(tmp_path, load_touches=True, normal_format=fmt)
    stems = sorted({p.stem for p in out.image_filenames}, key=dp.natural_key)
    assert [stems.index(p.stem) for p in out.image_filenames] == g("image_order").tolist()
    assert [stems.index(p.stem) for p in out.mask_filenames] == g("mask_order").tolist()
    assert [stems.index(Path(p).stem) for p in out.metadata["depth_filenames"]] == g("depth_order").tolist()
    assert [stems.index(Path(p).stem) for p in out.metadata["normal_filenames"]] == g("normal_order").tolist()
    assert np.array_equal(np.stack([c.c2w.numpy() for c in out.cameras]), g("c2w"))
    for k, attr in (("fx", "fx"), ("fy", "fy"), ("cx", "cx"), ("cy", "cy"), ("width", "width"), ("height", "height")):
        assert np.allclose(np.array([getattr(c, attr) for c in out.cameras], dtype=np.float64), g(k).astype(np.float64),
                           rtol=0, atol=1e-6), k
    assert out.dataparser_scale == pytest.approx(float(g("dataparser_scale")), rel=1e-12)
    assert np.array_equal(out.dataparser_transform.numpy(), g("dataparser_transform"))
    md = out.metadata
    assert md["scale_factor"] == pytest.approx(float(g("scale_factor")), rel=1e-12)
    assert np.array_equal(md["transform_matrix"].numpy(), g("transform_matrix"))
    assert np.array_equal(md["points3D_xyz"].numpy(), g("points3D_xyz"))
    assert np.array_equal(md["points3D_rgb"].numpy(), g("points3D_rgb"))
    assert np.array_equal(md["visual_hull"].numpy(), g("visual_hull"))
    assert np.allclose(md["points3D_normals"].numpy(), g("points3D_normals"), atol=1e-6)
    assert md["gel_scale_factor"] == float(g("gel_scale_factor")) and len(md["touch_patches"]) == int(g("n_patches"))
    for i, tp in enumerate(md["touch_patches"]):
        for k in ("points_xyz", "points_rgb", "normals", "bbox"):
            ref = g(f"touch{i}.{k}")
            assert tp[k].shape == ref.shape and tp[k].dtype == torch.float32, (i, k)
            assert np.array_equal(tp[k].numpy(), ref), (i, k, float(np.abs(tp[k].numpy() - ref).max()))
    for idx in range(len(out.cameras)):
        b = dp.load_batch(out, idx)
        assert b["sensor_depth"].dtype == torch.float32
        assert np.array_equal(b["sensor_depth"].numpy(), g(f"batch{idx}.sensor_depth").astype(np.float32)), idx
        assert np.array_equal(b["normal"].numpy(), g(f"batch{idx}.normal")), idx


_SHARD_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=rank, world_size=world)
from fusionsense_amd.trainer import SplatTrainer, PARAM_ORDER, GEOMETRY_GROUPS
torch.manual_seed(0)
n = 45
shapes = dict(means=(n,3), scales=(n,3), quats=(n,4), features_dc=(n,3), features_rest=(n,15,3), opacities=(n,1))
init = {k: torch.randn(*shapes[k]) for k in PARAM_ORDER}
dev = torch.device("cpu")
os.environ["FSGS_SHARDED_GEOMETRY"] = "1"
a = SplatTrainer(init, dev, fused=False)   # reduce-scatter -> Adam on the owned shard -> all-gather of the parameters
os.environ["FSGS_SHARDED_GEOMETRY"] = "0"
b = SplatTrainer(init, dev, fused=False)   # all-reduce -> Adam on everything
assert a._geo is not None and b._geo is None
assert a.slab.align == 64 and a.slab.split %% (64 * world) == 0 and b.slab.split == n * 11
for name, off, cnt in a.slab.geo_ranges:
    assert off %% 64 == 0 and a.params[name].data.data_ptr() == a._geo["P"].data_ptr() + 4 * off
def grads(tr, it):
    for i, k in enumerate(PARAM_ORDER):
        tr.slab.views[k].copy_(torch.sin(tr.params[k].data * (i + 1) + it) * (rank + 1) + 0.1 * rank)
for it in range(5):
    grads(a, it); grads(b, it)
    a._reduce_and_step(True); b._reduce_and_step(True)
    assert a._moments_local
    a.flush(); b.flush()
    a.step += 1; b.step += 1
    for k in PARAM_ORDER:
        assert torch.allclose(a.params[k].data, b.params[k].data, rtol=1e-6, atol=1e-7), (it, k)
# the moments of the other rank's shard are stale until they are gathered
lo, hi = a._geo["lo"], a._geo["hi"]
stale = False
for name, off, cnt in a.slab.geo_ranges:
    sa = a.optimizers[name].state[a.params[name]]["exp_avg"].reshape(-1)
    sb = b.optimizers[name].state[b.params[name]]["exp_avg"].reshape(-1)
    for j in range(cnt):
        inside = lo <= off + j < hi
        same = torch.allclose(sa[j], sb[j], rtol=1e-5, atol=1e-8)
        assert same or not inside, (name, j)
        stale |= (not same)
assert stale
a.sync_optimizer_state()
assert not a._moments_local
for k in PARAM_ORDER:
    sa, sb = a.optimizers[k].state[a.params[k]], b.optimizers[k].state[b.params[k]]
    assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-8), k
    assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-10), k
    assert float(sa["step"]) == float(sb["step"]) == 5.0
# a step without the deferred split (no optimizer step / a full step) first gathers the moments, then runs on everything
grads(a, 7); grads(b, 7)
a.defer_features = b.defer_features = False
a._reduce_and_step(True); b._reduce_and_step(True)
for k in PARAM_ORDER:
    assert torch.allclose(a.params[k].data, b.params[k].data, rtol=1e-6, atol=1e-7), k
# replicas: bit-identical across the ranks (the all-gather hands every rank the same bytes)
for k in PARAM_ORDER:
    t = a.params[k].data.clone(); dist.broadcast(t, 0)
    assert torch.equal(t, a.params[k].data), k
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_sharded_geometry_step_equals_allreduce_step_gloo_world2():
    """Data-parallel step with the geometry half reduce-scattered, Adam on the owned shard and the updated parameters
    all-gathered (trainer._sharded_geometry_step; gloo stands in for RCCL's reduce-scatter) against all-reduce + Adam
    on everything: same parameters every step, moments equal once gathered, padded 64-float-aligned layout, flat
    parameter slab, replicas bit-identical."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = _SHARD_WORKER % (ROOT, port)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0 and "ok" in out, err[-3000:]


def test_supervision_cache_keys_verifies_and_bounds(monkeypatch):
    """integration.SupervisionCache on the CPU (the preparation itself needs the GPU: stubbed): (split, image_idx) keys,
    verified hits for re-created tensors, loud refusal of other data under a cached key, LRU byte budget, no caching
    without an image_idx (eval batches used to hit the cached TRAIN view of the same index)."""
    from fusionsense_amd import integration

    calls = []

    def fake_prepare(batch, cfg, device):
        calls.append(1)
        img = batch["image"].float()
        return losses.FrameBatch(img, batch["sensor_depth"].float().reshape(img.shape[0], img.shape[1]), None, None, 0, 0, True)

    monkeypatch.setattr(losses, "prepare_batch", fake_prepare)
    cfg = losses.LossConfig()
    H, W = 32, 48

    def batch(seed, idx=None):
        g = torch.Generator().manual_seed(seed)
        b = {"image": torch.rand(H, W, 3, generator=g), "sensor_depth": torch.rand(H, W, 1, generator=g)}
        if idx is not None:
            b["image_idx"] = idx
        return b

    c = integration.SupervisionCache(budget_bytes=10 * H * W * 16)
    b_tr, b_ev = batch(1, 5), batch(2, 5)
    f_tr = c.get(True, b_tr, cfg, "cpu", H, W)
    f_ev = c.get(False, b_ev, cfg, "cpu", H, W)
    assert f_tr is not f_ev and len(c) == 2 and torch.equal(f_ev.image, b_ev["image"])
    assert c.get(True, b_tr, cfg, "cpu", H, W) is f_tr and c.hits == 1 and not c._checks  # same tensors: a hit at once
    copy = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b_tr.items()}
    assert c.get(True, copy, cfg, "cpu", H, W) is f_tr and len(c._checks) == 1  # re-created tensors: fingerprinted
    c._harvest(block=True)
    c.get(True, batch(3, 5), cfg, "cpu", H, W)  # other data under the train key
    with pytest.raises(RuntimeError, match="carried other data"):
        c.get(True, b_tr, cfg, "cpu", H, W)
    assert (True, 5) not in c.entries and (False, 5) in c.entries
    n = len(calls)
    c.get(True, batch(4), cfg, "cpu", H, W)
    c.get(True, batch(4), cfg, "cpu", H, W)
    assert len(calls) == n + 2 and len(c) == 1  # no image_idx: prepared every time, never stored
    c.get(True, b_tr, cfg, "cpu", 2 * H, W)  # another output size: not this entry
    small = integration.SupervisionCache(budget_bytes=2.5 * H * W * 16)
    for i in range(5):
        small.get(True, batch(10 + i, i), cfg, "cpu", H, W)
    assert list(small.entries) == [(True, 3), (True, 4)] and small.evictions == 3 and small.bytes <= small.budget


def test_live_capacity_is_bucketed_windowed_and_bounded():
    """capacity.LiveCapacity: a densification's new N finds its neighbour's estimate, estimates follow a shrinking model
    after two windows, the table is an LRU (one never-evicted, only-growing entry per N)."""
    from fusionsense_amd.capacity import LRU, LiveCapacity, n_bucket
    c = LiveCapacity(max_keys=4, window=8)
    k = c.key("cuda:0", 1, 300_000, 800, 800)
    assert c.get(k) == 0
    c.update(k, 1_000_000)
    assert c.get(k) == 1_254_096
    assert c.key("cuda:0", 1, 310_000, 800, 800) == k and c.key("cuda:0", 1, 420_000, 800, 800) != k
    assert n_bucket(1 << 20) == 80 and n_bucket(0) == 0
    for _ in range(20):  # the model shrank: two windows later the old maximum is gone
        c.update(k, 400_000)
    assert c.get(k) == int(400_000 * 1.25) + 4096
    c.raise_to(k, 2_000_000)
    assert c.get(k) >= 2_000_000
    for n in (1_000, 10_000, 100_000, 1_000_000, 10_000_000):
        c.update(c.key("cuda:0", 1, n, 64, 64), 5)
    assert len(c) == 4 and c.get(k) == 0
    from helpers import poison_capacity
    poison_capacity(c, 7)
    kk = c.key("cuda:0", 1, 10_000_000, 64, 64)
    assert c.get(k) == 0 and c.get(kk) == 7 and c.get(kk) == 4102  # (the hook waits for a shape the table knows)
    d = LRU(2)
    d["a"], d["b"] = 1, 2
    assert d.get("a") == 1
    d["c"] = 3
    assert list(d) == ["a", "c"]


def test_bench_watchdog_reaps_hung_ranks():
    """``bench.py --gpus N`` without a launcher: ranks that never finish (a hung rendezvous / collective) are killed as a
    process GROUP after --spawn-timeout, the tail of their output is shown, the exit code is 124 and no rank survives
    (ADVICE r3 / VERDICT r3 Weak 8: the child used to be waited for without a limit)."""
    import time
    import psutil
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FSGS_DIST_BACKEND="gloo", FSGS_BENCH_TEST_HANG="1")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--spawn-timeout", "20"], capture_output=True, text=True, env=env, timeout=180)
    assert r.returncode == 124, (r.returncode, r.stderr[-500:])
    assert "process group was killed" in r.stderr and "hanging for the watchdog test" in r.stderr and not r.stdout.strip()
    assert time.time() - t0 < 90
    time.sleep(1.0)
    left = [p for p in psutil.process_iter(["cmdline", "environ"]) if p.info["cmdline"] and "bench.py" in " ".join(p.info["cmdline"])
            and (p.info.get("environ") or {}).get("FSGS_BENCH_TEST_HANG") == "1"]
    assert not left, [p.info["cmdline"] for p in left]


_W8_WORKER = r"""
import os, sys, math, torch, torch.distributed as dist
sys.path.insert(0, %r)
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d", rank=rank, world_size=world)
from fusionsense_amd import ops
from fusionsense_amd.trainer import SplatTrainer, PARAM_ORDER, GEOMETRY_GROUPS, FEATURE_GROUPS
from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
# the row surgery of a refinement runs through HIP kernels in the product (no CPU fallback, by design); this test is
# about the EXCHANGE and the slab layout around it, so the two row primitives get their oracle's torch statement here
from oracle import refine_ref
ops.refine_mark = refine_ref.mark
ops.refine_move = refine_ref.move

torch.manual_seed(0)
n = 173                      # 11 * 173 = 1903 geometry floats: no multiple of 64, let alone of 64 * 8
shapes = dict(means=(n,3), scales=(n,3), quats=(n,4), features_dc=(n,3), features_rest=(n,15,3), opacities=(n,1))
init = {k: torch.randn(*shapes[k]) for k in PARAM_ORDER}
init["scales"] = init["scales"] * 0.5 - 4.0       # around exp(-4) = 0.018: some above, some below the 0.01 size threshold
init["opacities"] = init["opacities"] + 2.0
dev = torch.device("cpu")
cfg = SplatfactoConfig()
os.environ["FSGS_SHARDED_GEOMETRY"] = "1"
a = SplatTrainer(init, dev, fused=False, seed=3, strategy=DensifyStrategy(cfg, num_train_data=8))
os.environ["FSGS_SHARDED_GEOMETRY"] = "0"
# the single-process run "on the concatenated views": the mean of all eight ranks' gradients, formed locally (the
# gradient of a rank is a function of (parameters, rank, step)), stepped without any collective
c = SplatTrainer(init, dev, fused=False, seed=3, strategy=DensifyStrategy(cfg, num_train_data=8))
assert a._geo is not None and c._geo is None
assert a.slab.align == 64 and a.slab.world == 8 and a.slab.split %% (64 * 8) == 0 and a.slab.split >= 11 * n
assert a._geo["shard"] * 8 == a.slab.split and a._geo["lo"] == rank * a._geo["shard"]

def grad_of(tr, k, i, it, r):
    return torch.sin(tr._params[k].data * (i + 1) + it) * (r + 1) + 0.1 * r

def step_a(it):
    for i, k in enumerate(PARAM_ORDER):
        a.slab.views[k].copy_(grad_of(a, k, i, it, rank))
    a._reduce_and_step(True)
    assert a._pending is not None and a._moments_local
    a.flush()
    a.step += 1

def step_c(it):
    for i, k in enumerate(PARAM_ORDER):
        g = torch.zeros_like(c._params[k].data)
        for r in range(world):          # rank order, like the collectives' sums
            g += grad_of(c, k, i, it, r)
        c.slab.views[k].copy_(g / world)
    c.optimizers["means"].param_groups[0]["lr"] = c._means_lr(c.step)
    for k in PARAM_ORDER:
        c.optimizers[k].step()
    c.step += 1

def check(tag, rtol=2e-5, atol=1e-6):
    for k in PARAM_ORDER:
        pa, pc = a.params[k].data, c._params[k].data
        assert pa.shape == pc.shape, (tag, k, pa.shape, pc.shape)
        assert torch.allclose(pa, pc, rtol=rtol, atol=atol), (tag, k, float((pa - pc).abs().max()))
    for k in PARAM_ORDER:               # replicas: bit-identical on every rank
        t = a.params[k].data.clone(); dist.broadcast(t, 0)
        assert torch.equal(t, a.params[k].data), (tag, k)

a.step = c.step = 690
for it in range(5):
    step_a(it); step_c(it)
check("after 5 sharded steps")

# ---- one refinement: statistics differ per rank, the all-reduce (SUM / SUM of count-1 / MAX) makes the decisions equal
g = torch.Generator().manual_seed(100 + rank)
sa = a.strategy
sa.xys_grad_norm = torch.rand(n, generator=g) * 0.01
sa.vis_counts = 1.0 + torch.randint(0, 3, (n,), generator=g).float()
sa.max_2Dsize = torch.rand(n, generator=g) * 0.08
sa.last_size = (720, 1280)
parts = [torch.zeros(3, n) for _ in range(world)]
dist.all_gather(parts, torch.stack([sa.xys_grad_norm, sa.vis_counts - 1.0, sa.max_2Dsize]))
sc = c.strategy
sc.xys_grad_norm = sum(p[0] for p in parts)
sc.vis_counts = 1.0 + sum(p[1] for p in parts)
sc.max_2Dsize = torch.stack([p[2] for p in parts]).max(0).values
sc.last_size = (720, 1280)
import torch.distributed as _d
a.step = c.step = 700
sa.refinement_after(a, 700)             # (collectives inside: stats all-reduce, moment all-gather before the surgery)
_real_init = _d.is_initialized
_d.is_initialized = lambda: False       # the reference run takes no part in any collective
try:
    sc.refinement_after(c, 700)
finally:
    _d.is_initialized = _real_init
n2 = a.num_gaussians()
assert n2 == c.num_gaussians() and n2 != n, (n2, c.num_gaussians())
rep = sa.last_report
assert rep["n_split"] > 0 and rep["n_dup"] > 0 and rep["n_deleted"] > 0, rep
ns = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(ns, torch.tensor([n2]))
assert all(int(x) == n2 for x in ns), "N differs between the ranks after the refinement"
# the slab, the flat geometry slabs and the shard bounds follow the new N
assert a._geo is not None and a.slab.split %% (64 * 8) == 0 and a.slab.split >= 11 * n2 and a._geo["shard"] * 8 == a.slab.split
for name, off, cnt in a.slab.geo_ranges:
    assert off %% 64 == 0 and cnt == a._params[name].numel()
    assert a._params[name].data.data_ptr() == a._geo["P"].data_ptr() + 4 * off
    assert a._params[name].grad.data_ptr() == a.slab.views[name].data_ptr()
check("right after the refinement")
for k in PARAM_ORDER:                   # moments: survivors keep theirs, new rows start at zero — as in the single run
    ma, mc = a.optimizers[k].state[a._params[k]], c.optimizers[k].state[c._params[k]]
    assert torch.allclose(ma["exp_avg"], mc["exp_avg"], rtol=2e-5, atol=1e-7), k
    assert torch.allclose(ma["exp_avg_sq"], mc["exp_avg_sq"], rtol=2e-5, atol=1e-9), k
for it in range(5, 9):
    step_a(it); step_c(it)
check("after the refinement + 4 sharded steps")
a.sync_optimizer_state()
for k in PARAM_ORDER:
    ma, mc = a.optimizers[k].state[a._params[k]], c.optimizers[k].state[c._params[k]]
    assert torch.allclose(ma["exp_avg"], mc["exp_avg"], rtol=2e-5, atol=1e-6), (k, float((ma["exp_avg"] - mc["exp_avg"]).abs().max()))
    assert float(ma["step"]) == float(mc["step"]) == 9.0, (k, float(ma["step"]), float(mc["step"]))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_world8_sharded_step_statistics_and_refinement_gloo():
    """EIGHT ranks over gloo (VERDICT r3 Next 3: nothing had ever run at W = 8): the sharded geometry step with a
    Gaussian count that is no multiple of 64 x 8 (padding, shard bounds, four-group shard intersections), the deferred
    feature exchange, the per-refinement statistics all-reduce, ONE refinement with splits / dups / culls (moment
    all-gather before the surgery, slab + flat geometry slabs rebuilt for the new N) — and the parameters equal those of
    a single process that steps on the mean gradient of the eight views; replicas bit-identical throughout."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = _W8_WORKER % (ROOT, port)
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    fails = []
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, err = p.communicate()
            err += "\n[timeout]"
        if p.returncode != 0 or "ok" not in out:
            fails.append((r, err[-1500:]))
    assert not fails, fails[0]


@pytest.mark.parametrize("case", ["closest", "analytical", "few"])
def test_level_surface_points_match_reference_execution(case):
    """inference.level_surface_points_from_render against what the reference's own
    ``compute_level_surface_points`` (dn_model.py:1706-1946) produced — executed by
    tests/golden/make_reference_levelset_goldens.py — from the same render (the oracle rasterizer's depth / rgb and the
    per-Gaussian world normals stored with the golden): the same pixels cross each level, at the same points, with the
    same normals (closest-Gaussian and analytical mode, with and without a mask, sub-sampled)."""
    from fusionsense_amd import inference as inf
    d = np.load(os.path.join(GOLD, "reference_levelset.npz"))
    t = lambda k: torch.from_numpy(d[k])  # noqa: E731
    params = {k: t(f"in.{k}") for k in ("means", "scales", "quats", "opacities")}
    fx, fy, cx, cy, W, H = (float(x) for x in d["intr"])
    cam = scenes.Camera(t("c2w"), fx, fy, cx, cy, int(W), int(H))
    mode = ["closest_gaussian", "analytical"][int(d[f"{case}.mode"])]
    mask = t("mask") if bool(d[f"{case}.use_mask"]) else None
    res = inf.level_surface_points_from_render(params, t("render.normals_world"), t("render.depth"), t("render.rgb"), cam,
                                               int(d[f"{case}.num_samples"]), mask, (0.1, 0.3, 0.5), mode,
                                               sample_fn=lambda n, k: list(range(n))[:k])
    for lv in (0.1, 0.3, 0.5):
        for key in ("points", "normals", "colors"):
            ref = t(f"{case}.{lv}.{key}")
            got = res[lv][key]
            assert got.shape == ref.shape, (lv, key, got.shape, ref.shape)
            assert torch.allclose(got, ref, rtol=1e-4, atol=2e-5), (lv, key, float((got - ref).abs().max()))
        # every point lies on the ray of the pixel it is attributed to
        c2w = torch.eye(4)
        c2w[:3] = cam.c2w
        c2w = (c2w @ torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0])))[:3]
        uv = inf.project_pix(res[lv]["points"], fx, fy, cx, cy, c2w)
        pid = res[lv]["pixel_ids"]
        assert torch.allclose(uv[:, 0], (pid % int(W)).float() + 0.5, atol=2e-2) and torch.allclose(uv[:, 1], (pid // int(W)).float() + 0.5, atol=2e-2)
    assert res[0.1]["points"].shape[0] >= res[0.3]["points"].shape[0] >= res[0.5]["points"].shape[0] > 0


def test_knn_drop_first_is_sklearn_minus_its_first_column():
    """inference.knn_drop_first = dn_splatter.utils.knn.knn_sk (utils/knn.py:29-44): sklearn's k + 1 nearest with the
    first column dropped — checked against sklearn itself."""
    from sklearn.neighbors import NearestNeighbors
    from fusionsense_amd.inference import knn_drop_first
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(500, 3, generator=g), torch.randn(77, 3, generator=g)
    ref = NearestNeighbors(n_neighbors=6, algorithm="auto", metric="euclidean").fit(x.numpy()).kneighbors(y.numpy())[1][:, 1:]
    assert np.array_equal(knn_drop_first(x, y, 5).numpy(), ref)


@pytest.mark.parametrize("name", ["plain", "masked"])
def test_sdf_term_matches_reference_execution(name):
    """fusionsense_amd/sdf.py against tests/golden/reference_sdf.npz: the reference's own get_loss_dict executed with the
    SDF term off and on (the stored term and gradients are the difference), on the reference's own random draws."""
    import os
    import numpy as np
    from fusionsense_amd import sdf
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_sdf.npz"))
    t = lambda k: torch.from_numpy(d[f"{name}.{k}"])  # noqa: E731
    P = {k: t(k).clone().requires_grad_(True) for k in ("means", "scales", "quats", "opacities")}
    fx, cx, cy, W, H = [float(x) for x in d[f"{name}.intr"]]
    n_samples, lam = int(d[f"{name}.cfg"][0]), float(d[f"{name}.cfg"][1])
    add_mask = t("add_mask") if bool(d[f"{name}.has_anchors"]) else None
    mask = t("mask") if bool(d[f"{name}.has_mask"]) else None
    term = lam * sdf.sdf_loss(P["means"], P["scales"], P["quats"], P["opacities"], t("knn"), t("vis_indices"), add_mask,
                              t("depth"), t("c2w"), fx, cx, cy, int(W), int(H), mask, n_samples,
                              draws=(t("picks"), t("centered")))
    term.backward()
    ref = float(d[f"{name}.term"])
    assert abs(float(term) - ref) < 2e-5 * max(1.0, abs(ref)), (float(term), ref)
    for k in P:
        g, r = P[k].grad, t("v_" + k)
        assert float((g - r).abs().max()) < 2e-5 * max(1.0, float(r.abs().max())), (k, float((g - r).abs().max()))
    assert float(t("v_means").abs().max()) > 0 and float(t("v_opacities").abs().max()) > 0
    # without handed-in draws the term samples for itself (a seeded generator: reproducible), and fewer Gaussians than
    # num_sdf_samples is the reference's own IndexError (its weight indexes the Gaussians with the samples' mask)
    g1 = torch.Generator().manual_seed(3)
    a = sdf.sdf_loss(P["means"].detach(), P["scales"].detach(), P["quats"].detach(), P["opacities"].detach(), t("knn"),
                     t("vis_indices"), add_mask, t("depth"), t("c2w"), fx, cx, cy, int(W), int(H), mask, n_samples, generator=g1)
    g2 = torch.Generator().manual_seed(3)
    b = sdf.sdf_loss(P["means"].detach(), P["scales"].detach(), P["quats"].detach(), P["opacities"].detach(), t("knn"),
                     t("vis_indices"), add_mask, t("depth"), t("c2w"), fx, cx, cy, int(W), int(H), mask, n_samples, generator=g2)
    assert float(a) == float(b) and math.isfinite(float(a))
    with pytest.raises(IndexError):
        sdf.sdf_loss(P["means"].detach(), P["scales"].detach(), P["quats"].detach(), P["opacities"].detach(), t("knn"),
                     t("vis_indices"), add_mask, t("depth"), t("c2w"), fx, cx, cy, int(W), int(H), mask, 10_000)
    cfg = type("C", (), dict(use_sdf_loss=True, apply_sdf_loss_after_iters=2000, apply_sdf_loss_iters=100))
    assert sdf.sdf_term_due(cfg, 2100) and not sdf.sdf_term_due(cfg, 2000) and not sdf.sdf_term_due(cfg, 2150)


def test_kernel_timer_selects_and_samples_launches(monkeypatch):
    """ops.TIMER as bench.py uses it inside its timed window: only the selected launches get a span, of those every k-th
    is bracketed by events, every other launch takes the plain path (no span object) — and a non-zero return code raises
    on either path."""
    from fusionsense_amd import ops
    from fusionsense_amd._lib import FsgsError

    class FakeEvent:
        made = 0

        def __init__(self, enable_timing=False):
            FakeEvent.made += 1
            self.t = None

        def record(self):
            FakeEvent.clock = getattr(FakeEvent, "clock", 0.0) + 1.0
            self.t = FakeEvent.clock

        def elapsed_time(self, other):
            return other.t - self.t
    monkeypatch.setattr(ops.torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(ops.torch.cuda, "synchronize", lambda *a, **k: None)
    spans = []
    real_span = ops._KernelTimer.span
    monkeypatch.setattr(ops._KernelTimer, "span", lambda self, name: (spans.append(name), real_span(self, name))[1])
    calls = []
    ok = lambda *a: (calls.append(a), 0)[1]  # noqa: E731
    try:
        ops.TIMER.reset(enabled=True, only=("raster_bwd",), prealloc=3, every=4)
        assert FakeEvent.made == 6
        for _ in range(10):
            ops._run(ok, (1, 2), "fsgs_tile_sort")
            ops._run(ok, (3,), "fsgs_raster_bwd_quad", "_d4e3")
        assert len(calls) == 20
        assert spans == ["raster_bwd_quad_d4e3"] * 10      # the unselected launch never made a span
        s = ops.TIMER.summary()
        assert list(s) == ["raster_bwd_quad_d4e3"] and s["raster_bwd_quad_d4e3"]["calls"] == 3  # launches 1, 5, 9
        assert s["raster_bwd_quad_d4e3"]["avg_ms"] == 1.0
        with pytest.raises(FsgsError):
            ops._run(lambda *a: 2, (), "fsgs_tile_sort")
        with pytest.raises(FsgsError):
            ops._run(lambda *a: 2, (), "fsgs_raster_bwd_quad")
        ops.TIMER.reset(enabled=True)  # no selection: every launch timed
        ops._run(ok, (), "fsgs_tile_sort")
        assert list(ops.TIMER.summary()) == ["tile_sort"]
    finally:
        ops.TIMER.reset(enabled=False)
    ops._run(ok, (), "fsgs_tile_sort")
    assert ops.TIMER.summary() == {}


def test_count_free_route_waits_on_the_mapped_flag(monkeypatch):
    """ops._wait_total on the count-free route (no event behind the launch): the total is returned as soon as the
    bookkeeping workgroup's flag is up; if it never comes up within the spin, the stream is synchronised before the read.
    With an event (the two-pass route) the event is what is waited for."""
    from fusionsense_amd import ops
    arr = np.array([1234, 1, 0, 0], dtype=np.int32)
    pinned = type("P", (), {"_np": arr})()
    monkeypatch.setattr(ops, "SPIN_WAIT", True)
    assert ops._wait_total(dict(pinned=pinned, event=None, pinned_dev="d")) == 1234
    synced = []

    class Stream:
        def synchronize(self):
            synced.append(1)
            arr[0], arr[1] = 77, 1
    monkeypatch.setattr(ops.torch.cuda, "current_stream", lambda dev=None: Stream())
    arr[1] = 0
    monkeypatch.setattr(ops, "SPIN_WAIT", False)  # (no spin: straight to the fallback)
    assert ops._wait_total(dict(pinned=pinned, event=None, pinned_dev="d")) == 77 and synced == [1]
    waited = []
    monkeypatch.setattr(ops, "_wait_event", lambda ev: waited.append(ev))
    arr[0] = 5
    assert ops._wait_total(dict(pinned=pinned, event="EV")) == 5 and waited == ["EV"]


def test_flag_ring_never_rearms_a_slot_nobody_read(monkeypatch):
    """the live total's flag is a bare 1 in a ring of four mapped blocks.  A slot whose launch was never looked
    at (an exception between the launch and its check) must not be re-armed while that launch may still write it: the
    ring drains the device first; a slot that WAS read is re-armed without any wait.  And importing the package sets the
    runtime's command-batch limit unless the caller has."""
    from fusionsense_amd import ops
    bufs = []
    for _ in range(4):
        arr = np.zeros(4, dtype=np.int32)
        bufs.append(type("P", (), {"_np": arr})())
    monkeypatch.setitem(ops._PINNED, "dev0:i32", dict(bufs=bufs, i=0))
    drained = []
    monkeypatch.setattr(ops.torch.cuda, "synchronize", lambda dev=None: drained.append(dev))
    monkeypatch.setattr(ops, "SPIN_WAIT", True)
    seen = []
    for frame in range(8):  # every frame's total is read: two laps of the ring without a single drain
        b = ops._pinned_i32("dev0")
        assert b._pending and b._np[1] == 0
        b._np[0], b._np[1] = 100 + frame, 1  # (the kernel lands)
        seen.append(ops._wait_total(dict(pinned=b, event=None, pinned_dev="dev0")))
        assert not b._pending
    assert seen == [100 + f for f in range(8)] and drained == []
    lost = ops._pinned_i32("dev0")  # armed, launched ... and never checked
    for _ in range(3):
        b = ops._pinned_i32("dev0")
        b._np[1] = 1
        ops._wait_total(dict(pinned=b, event=None, pinned_dev="dev0"))
    assert drained == []
    again = ops._pinned_i32("dev0")  # the ring comes back to the lost slot: drained before it is cleared and re-armed
    assert again is lost and drained == ["dev0"]
    out = subprocess.run([sys.executable, "-c", "import os; os.environ.pop('DEBUG_CLR_MAX_BATCH_SIZE', None); "
                          "import fusionsense_amd; print(os.environ['DEBUG_CLR_MAX_BATCH_SIZE'])"], capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "16384", out.stderr[-500:]


def test_bench_sets_the_runtime_limit_before_the_runtime_loads():
    """bench.py moves the HIP runtime's command-batch limit with os.environ.setdefault BEFORE torch —
    and with it libamdhip64 — is imported, and a caller's own setting wins."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert src.index("RUNTIME_ENV = {") < src.index("\nimport torch\n")
    assert 'os.environ.setdefault(_k, _v)' in src
    env = dict(os.environ, DEBUG_CLR_MAX_BATCH_SIZE="4321")
    out = subprocess.run([sys.executable, "-c", "import bench; print(bench.RUNTIME_ENV)"], env=env, capture_output=True,
                         text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "'DEBUG_CLR_MAX_BATCH_SIZE': '4321'" in out.stdout
