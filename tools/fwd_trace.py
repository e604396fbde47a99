"""Per-wave timeline of the wave-per-quadrant forward (diagnostic build):
    make -C fusionsense_amd/csrc OUT=../libfsgs_trace.so BUILD=build_trace EXTRA=-DFSGS_FWD_TRACE
    FSGS_LIB=$PWD/fusionsense_amd/libfsgs_trace.so FSGS_FWD_WALK=1 python tools/fwd_trace.py [config]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fusionsense_amd import _lib  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sys.argv = ["bench.py", "--config", str(cfg), "--no-cpu-baseline", "--no-dropin"] + (["--views", "8"] if cfg == 2 else [])
args = bench.parse()
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wl = bench.build_workload(args, dev)
tr, cams, targets = wl["trainer"], wl["cams"], wl["targets"]
lib = C.CDLL(_lib.LIB_PATH)
for s in range(6):
    tr.train_step(cams[s % len(cams)], targets[s % len(cams)])
torch.cuda.synchronize()
W, H = cams[0].width, cams[0].height
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
nb = ((n_tiles + 7) // 8) * 32 * 2
nb = min(nb, 1 << 17)
out = np.zeros((nb, 6), dtype=np.int64)
assert lib.fsgs_debug_fwd_trace(out.ctypes.data_as(C.POINTER(C.c_longlong)), nb) == 0
last = out[:, 0].max()
out[out[:, 0] < last - 30000] = 0  # (blocks that left at once keep the entries of earlier launches: 300 us window)
ran = out[:, 2] > 0
t0 = out[out[:, 0] > 0, 0].min()
st, en = (out[:, 0] - t0) / 100.0, (out[:, 1] - t0) / 100.0  # 100 MHz -> microseconds
print(json.dumps({"blocks": int(nb), "walked": int(ran.sum()), "launch_span_us": round(float(en[out[:, 1] > 0].max()), 2),
                  "records_total": int(out[ran, 2].sum()), "records_max": int(out[ran, 2].max()),
                  "list_max": int(out[ran, 3].max())}))
life = en - st
order = np.argsort(-out[:, 2])[:12]
print("longest walks: records, list, start_us, end_us, life_us, us/record")
for b in order:
    print(int(out[b, 2]), int(out[b, 3]), round(float(st[b]), 2), round(float(en[b]), 2), round(float(life[b]), 2),
          round(float(life[b] / max(out[b, 2], 1)), 4))
# when do waves end: histogram of end times
idx = np.where(ran)[0]
ends = np.sort(en[idx])
print("end-time percentiles (us):", [round(float(np.percentile(ends, p)), 1) for p in (10, 50, 90, 99, 100)])
print("start-time percentiles (us):", [round(float(np.percentile(st[idx], p)), 1) for p in (10, 50, 90, 99, 100)])
rec = out[idx, 2].astype(np.float64)
print("us per record, by walk length quartile:", [round(float((life[idx][rec <= np.percentile(rec, p)]).sum() / rec[rec <= np.percentile(rec, p)].sum()), 4) for p in (25, 50, 75, 100)])
# concurrency over time
grid = np.linspace(0, ends[-1], 24)
print("waves in flight:", [int(((st[idx] <= t) & (en[idx] > t)).sum()) for t in grid])
