"""Lane utilisation of the compositing backward (raster_bwd_live_kernel), from the instrumented build:
    make -C fusionsense_amd/csrc OUT=../libfsgs_stats.so BUILD=build_stats EXTRA=-DFSGS_BWD_STATS
    FSGS_LIB=$PWD/fusionsense_amd/libfsgs_stats.so python tools/bwd_lane_stats.py [config]
Prints, over a few training frames: executed steps, lanes with a list entry, lanes that pass the alpha test (of the 64 of
a step), share of merged (union-walk) segments, longest-row / union ratio."""
import ctypes as C
import json
import os
import sys

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
out = (C.c_ulonglong * 12)()
for s in range(4):
    tr.train_step(cams[s % len(cams)], targets[s % len(cams)])
torch.cuda.synchronize()
assert lib.fsgs_debug_bwd_stats(out, 1) == 0
frames = 8
for s in range(frames):
    tr.train_step(cams[s % len(cams)], targets[s % len(cams)])
torch.cuda.synchronize()
assert lib.fsgs_debug_bwd_stats(out, 0) == 0
it, ex, have, valid, seg, mer, lng, uni, hmax, qmax, hpairs, qpairs = [int(x) for x in out]
# (the tuner issues the launch twice on its measuring frames: ratios are unaffected)
print(json.dumps({
    "config": cfg, "frames": frames, "iterations_per_frame": it // frames, "executed_steps_per_frame": ex // frames,
    "executed_over_iterations": round(ex / max(it, 1), 4),
    "lanes_with_entry_per_step": round(have / max(ex, 1), 2), "lanes_valid_per_step": round(valid / max(ex, 1), 2),
    "lane_utilisation": round(valid / max(64 * ex, 1), 4),
    "segments_per_frame": seg // frames, "merged_segment_frac": round(mer / max(seg, 1), 4),
    "longest_row_over_union": round(lng / max(uni, 1), 4), "mean_longest_row": round(lng / max(seg, 1), 2),
    "mean_union": round(uni / max(seg, 1), 2),
    # what finer pixel groups would need (lower bounds: exact validity instead of a conservative reach test):
    # steps if every 4x2 half-row / 2x2 quad walked its own list, relative to today's executed steps, and the
    # (group, record) pairs that would each issue 15 atomics, relative to today's (row, record) pairs
    "steps_with_half_rows_over_now": round(hmax / max(ex, 1), 4), "steps_with_quads_over_now": round(qmax / max(ex, 1), 4),
    "half_row_pairs_over_row_pairs": round(hpairs / max(have / 16.0, 1), 4),
    "quad_pairs_over_row_pairs": round(qpairs / max(have / 16.0, 1), 4)}))
