#!/usr/bin/env python3
"""Pretty-print a bench.py JSON line."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = d.pop("kernels_ms", {})
cfg = d.pop("config")
print(json.dumps(d, indent=1))
print({a: cfg[a] for a in cfg if a != "workload"})
tot = 0.0
for n, v in k.items():
    per_step = v["avg_ms"]  # one launch of each kind per step on the fused path
    tot += per_step
    print(f"{n:24s} avg {v['avg_ms']:8.4f} ms ({v['calls']} launches timed)")
print(f"sum of fsgs kernels per step: {tot:.3f} ms of {d['ms_per_step']} ms")
