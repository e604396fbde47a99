#!/usr/bin/env python3
"""Tabulate the rocprofv3 --pmc passes of tools/valu_rate.sh: per kernel of tools/valu_rate.hip the counters per
chip-wide launch, and what they mean per wave instruction.

    python tools/valu_rate_pmc.py gpurun_out/valu_rate

chain<OP, EXEC> launches: only the chip-wide ones (2048 workgroups of 256) are kept; a launch issues
waves x reps x 64 instructions of the class (reps = 256 under --quick).  atom<PAT, MODE> launches: 8192 waves x it
instructions (it = 128; 32 for the 64-lines pattern and the one-address pattern)."""
import collections
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
_src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "valu_rate.hip")).read()
OPS = re.findall(r'"([^"]*)"', _src[_src.index("kOpName[N_OPS] = {"):_src.index("};", _src.index("kOpName[N_OPS] = {"))])
EXEC = ["full", "low32", "even", "low16", "one", "EMPTY", "row15"]
PATS = ["atomic 15 lanes/row -> line", "atomic 16 lanes/row -> line", "atomic 64 lanes -> 64 lines", "atomic 64 lanes -> 256 B",
        "atomic 64 lanes -> 1 address", "store 16 B/lane", "store 4 B/lane", "store row15 pattern", "load 16 B/lane"]
MODES = ["stream", "hash 32 MB", "hash 8 MB", "hash 256 KB"]

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        grid = int(r.get("Grid_Size", 0) or 0)
        if k.startswith("void chain") or k.startswith("chain"):
            if grid != 2048 * 256:
                continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))

REPS_QUICK, UNROLL, WAVES = 256, 64, 8192
print("== chain kernels (chip-wide launch: %d waves x %d x %d instructions) ==" % (WAVES, REPS_QUICK, UNROLL))
for k in sorted(acc):
    m = re.search(r"chain<(\d+), (\d+)>", k)
    if not m:
        continue
    op, ex = int(m.group(1)), int(m.group(2))
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    n = WAVES * REPS_QUICK * UNROLL
    if op == OPS.index("global_atomic_add_f32 (issue)"):
        n //= 16
    line = "%-32s %-6s" % (OPS[op], EXEC[ex])
    for c in sorted(d):
        line += "  %s=%.4g (%.3f/inst)" % (c, d[c], d[c] / n)
    print(line)
print()
print("== atom kernels (8192 waves; KB counters as reported, per wave instruction in bytes) ==")
for k in sorted(acc):
    m = re.search(r"atom<(\d+), (\d+)>", k)
    if not m:
        continue
    pat, mode = int(m.group(1)), int(m.group(2))
    it = 128
    if pat == 2 or (pat == 4 and mode == 3):
        it = 32
    if pat in (5, 6, 8):
        it = 512
    n = WAVES * it
    d = {c: sum(v) / len(v) for c, v in acc[k].items()}
    line = "%-30s %-11s insts=%d" % (PATS[pat], MODES[mode], n)
    for c in sorted(d):
        per = d[c] / n * (1024.0 if c.endswith("_SIZE") else 1.0)
        line += "  %s=%.5g (%.2f %s/inst)" % (c, d[c], per, "B" if c.endswith("_SIZE") else "")
    print(line)
