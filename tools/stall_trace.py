#!/usr/bin/env python3
"""Slowest steps of a bench.py run in a rocprofv3 --kernel-trace CSV: the trace is cut into steps at the compositing
backward's launches; prints the longest intervals (under `cap_ms`) with their launches (start offset, duration, name).
Usage: tools/stall_trace.py trace_kernel_trace.csv [n=2] [cap_ms=60]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cap = float(sys.argv[3]) if len(sys.argv) > 3 else 60.0
marks = [i for i, r in enumerate(rows) if "raster_bwd_live_kernel" in r[2]]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    dt = (rows[b][0] - rows[a][0]) / 1e6
    if dt < cap:
        steps.append((dt, a, b))
print("steps", len(steps), "median ms", sorted(s[0] for s in steps)[len(steps) // 2])
ranked = sorted(steps, reverse=True)
for dt, a, b in ranked[:n] + [ranked[len(ranked) // 2]]:  # ... and the median step
    print(f"--- step of {dt:.3f} ms, backward #{marks.index(a)} of {len(marks)}")
    t0, end = rows[a][0], rows[a][0]
    for s, e, name in rows[a:b + 1]:
        short = name.replace("void ", "").replace("fsgs::", "").replace("at::native::", "")[:48]
        print(f"   +{(s - t0) / 1e3:9.1f} us  gap {max(0, s - end) / 1e3:8.1f}  dur {(e - s) / 1e3:7.1f}  {short}")
        end = max(end, e)
