"""2 ranks (gloo, sharing this GPU): data-parallel training WITH densification; replicas must stay bit-identical."""
import sys, os, math, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fusionsense_amd import scenes
from fusionsense_amd.trainer import SplatTrainer
from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
N0, RES, V, STEPS = 40_000, 256, 8, int(sys.argv[1]) if len(sys.argv) > 1 else 600
cams = scenes.hemisphere_cameras(V, width=RES, height=RES, focal=1111.11 * RES / 800, seed=0)
cfg = SplatfactoConfig(warmup_length=100, refine_every=50, reset_alpha_every=6, stop_split_at=500)
tr = SplatTrainer(scenes.lego_like_scene(N0, seed=0), dev, sh_degree=3, strategy=DensifyStrategy(cfg, num_train_data=V), seed=0)
gt = SplatTrainer(scenes.lego_like_scene(N0, seed=1), dev, sh_degree=3)
with torch.no_grad():
    tg = [{k: gt.forward(c)[k].detach().clone() for k in ("rgb", "depth", "normal")} for c in cams]
sizes = set()
for s in range(STEPS):
    v = (s * world + rank) % V
    loss, _ = tr.train_step(cams[v], tg[v])
    sizes.add(tr.num_gaussians())
tr.flush()
n = torch.tensor([tr.num_gaussians()], device=dev); n0 = n.clone(); dist.broadcast(n0, 0)
assert int(n) == int(n0), (rank, int(n), int(n0))
for k, p in tr.params.items():
    ref = p.data.clone(); dist.broadcast(ref, 0)
    assert torch.equal(ref, p.data), (rank, k)
assert math.isfinite(float(loss))
print(f"rank {rank}: ok, N {N0} -> {tr.num_gaussians()} ({len(sizes)} sizes), loss {float(loss):.4f}, deferred={tr.defer_features} factored={tr.factored_features} sharded={tr._geo is not None} cut_frames={tr.cut_frames} cut_redone={tr.cut_redone}", flush=True)
dist.destroy_process_group()
