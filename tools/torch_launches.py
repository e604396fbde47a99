"""Which torch (non-libfsgs) kernels a training step of a configuration launches, and from where: runs a few steps under
torch.profiler and prints the aten ops with CUDA time together with their Python call sites.
    python tools/torch_launches.py [config]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
sys.argv = ["bench.py", "--config", str(cfg), "--no-cpu-baseline", "--no-dropin"]
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wl = bench.build_workload(args, dev)
tr, cams, targets = wl["trainer"], wl["cams"], wl["targets"]
if wl.get("start_step"):
    tr.step = wl["start_step"]
for s in range(8):
    tr.train_step(cams[s % len(cams)], targets[s % len(cams)])
torch.cuda.synchronize()
steps = 20
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for s in range(steps):
        tr.train_step(cams[s % len(cams)], targets[s % len(cams)])
    torch.cuda.synchronize()
print(f"config {cfg}: aten ops with device time over {steps} steps")
for e in sorted(prof.key_averages(group_by_stack_n=6), key=lambda e: -e.count):
    if e.key.startswith("aten::") and (getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)) > 0:
        stack = [f for f in e.stack if "fusionsense_amd" in f or "bench.py" in f][:3]
        print(f"{e.key:32s} calls {e.count:4d}  device {getattr(e, 'device_time_total', getattr(e, 'cuda_time_total', 0)) / max(e.count, 1):7.1f} us  {e.input_shapes}  <- {stack}")
