"""Host-side profile of config #3's callback steps (add_touch_patch, refinement + hull / touch pruning)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = ["bench.py", "--config", "3", "--no-cpu-baseline", "--no-dropin"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wl = bench.build_workload(args, dev)
tr, cams, targets = wl["trainer"], wl["cams"], wl["targets"]
tr.step = 990
st = tr.strategy
pr = cProfile.Profile()
times = []
for s in range(220):
    v = s % len(cams)
    cb = (tr.step == st.add_touch_at) or (tr.step % st.cfg.refine_every == 0)
    if cb:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pr.enable()
    tr.train_step(cams[v], targets[v])
    if cb:
        torch.cuda.synchronize()
        pr.disable()
        times.append((tr.step - 1, round((time.perf_counter() - t0) * 1e3, 2)))
print("callback steps (step, ms incl. the step itself):", times)
for key in ("cumulative", "tottime"):
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(40)
    print(buf.getvalue()[:7000])
