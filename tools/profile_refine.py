"""Host-side timing of config #3's callback steps (add_touch_patch, refinement + hull / touch pruning): wall time of the
callbacks' sections, by wrapping the functions they are made of (no profiler: its overhead is of the order of the sections)."""
import os
import sys
import time
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fusionsense_amd import ops  # noqa: E402

sys.argv = ["bench.py", "--config", "3", "--no-cpu-baseline", "--no-dropin"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wl = bench.build_workload(args, dev)
tr, cams, targets = wl["trainer"], wl["cams"], wl["targets"]
tr.step = 990
st = tr.strategy
acc = defaultdict(float)
calls = defaultdict(int)


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += (time.perf_counter() - t0) * 1e3
            calls[label] += 1
    setattr(obj, name, w)


for name in ("refine_mark", "refine_move", "nearest_point_words", "nearest_point"):
    timed(ops, name)
for name in ("refinement_after", "hull_and_touch_pruning", "add_touch_patch", "_rebuild_marked", "cull_gaussians", "_rules",
             "_touch_static_rows"):
    timed(st, name)
timed(tr.slab, "rebuild", "slab.rebuild")
timed(tr, "flush", "trainer.flush")
timed(torch, "randn", "torch.randn")
timed(torch, "empty", "torch.empty")
rows = []
for s in range(320):
    v = s % len(cams)
    cb = (tr.step == st.add_touch_at) or (tr.step % st.cfg.refine_every == 0)
    if cb:
        acc.clear(); calls.clear()
        t0 = time.perf_counter()
    tr.train_step(cams[v], targets[v])
    if cb:
        t1 = time.perf_counter()
        rows.append((tr.step - 1, round((t1 - t0) * 1e3, 3), {k: (round(v_, 3), calls[k]) for k, v_ in sorted(acc.items())}))
for r in rows:
    print(r[0], "step incl. callbacks (host, no sync added):", r[1], "ms")
    for k, (ms, n) in r[2].items():
        print(f"    {k:28s} {ms:8.3f} ms  x{n}")
