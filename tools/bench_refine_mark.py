"""Device time of fsgs_refine_mark's variants on config #3's shapes (75 k rows, hull of 20 k points in a 0.1 ball)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fusionsense_amd import ops, scenes, touch
from fusionsense_amd.splatfacto import DensifyStrategy, SplatfactoConfig

dev = torch.device("cuda:0")
params, _ = scenes.fusionsense_like_scene(seed=0, n_hull=20000, n_bg=55000)
P = {k: v.to(dev) for k, v in params.items()}
hull = P["means"][:20000].clone()
n = P["means"].shape[0]
st = DensifyStrategy(SplatfactoConfig(), num_train_data=9)
center = [float(x) for x in hull.mean(0).tolist()]
grid = touch.hull_grid(hull, 0.02)
g = torch.Generator().manual_seed(0)
add = (torch.rand(n, generator=g) < 0.1).to(dev)


def rules():
    r = st._rules(1100, False)
    r.hull_center[0], r.hull_center[1], r.hull_center[2] = center
    r.hull_close, r.hull_lo, r.hull_hi = 0.2, 0.005, 0.02
    return r


def run(label, close=0.2, **kw):
    ts = []
    for it in range(6):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        r = rules()
        r.hull_close = close
        m = ops.refine_mark(r, P["scales"], P["opacities"], means=P["means"], add_mask=add, **kw)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    print(f"{label:34s} us: " + " ".join(f"{t:7.1f}" for t in ts), " n_keep", m.n_keep)


run("plain cull")
run("hull grid", hull_grid=grid)
run("hull grid, no row close", close=0.0, hull_grid=grid)
far = touch.hull_grid(hull + 0.5, 0.02)
run("hull grid of a far hull (no points)", hull_grid=far)
cnt = (grid["cells"][1:] - grid["cells"][:-1])
print("grid dims", grid["dims"], "points per non-empty cell", float(cnt[cnt > 0].float().mean()))
words = ops.nearest_point_words(P["means"], hull, center, 0.2)
run("hull words (search not included)", hull_words=words)
for it in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    words = ops.nearest_point_words(P["means"], hull, center, 0.2)
    b.record(); torch.cuda.synchronize()
    print("nearest_point_words us", a.elapsed_time(b) * 1e3)
