"""Instruction mix of a kernel's loop bodies from hipcc's --save-temps assembly:
    python tools/isa_mix.py <file.s> <substring of the mangled kernel name> [--json out.json [key]]
Splits the function into basic blocks at labels / branches, keeps the blocks that sit inside a loop (a backward branch
targets them or a later block branches back over them) and prints, per loop body, the instruction classes."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
json_out = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
json_key = sys.argv[sys.argv.index("--json") + 2] if json_out and len(sys.argv) > sys.argv.index("--json") + 2 else key
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (key in l and re.match(r"^_Z\S+:\s", l)))
end = next(i for i in range(start + 1, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
body = lines[start + 1:end]
blocks, cur, name = [], [], "entry"
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    cur.append(t.split(";")[0].strip())
blocks.append((name, cur))
index = {n: i for i, (n, _) in enumerate(blocks)}


def klass(ins):
    op = ins.split()[0]
    if "_dpp" in ins or "row_" in ins or "quad_perm" in ins:
        return "valu_dpp"
    if op.startswith("v_permlane"):
        return "valu_permlane"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(("v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_log")):
        return "valu_transcendental"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"):
        return "valu_cmp_select"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_atomic") or op.startswith("flat_atomic"):
        return "atomic"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    return "salu"


# Issue classes as MEASURED on gfx950 (tools/valu_rate.hip, profiles/r6_valu_rate.txt): cycles a SIMD needs per wave64
# instruction with >= 2 waves resident.  "simple": plain fp32 / integer ops that two waves can issue side by side (2.2 cycles
# in streams of their own kind, 4.4 for a lone wave or next to DPP / transcendental work); "full": everything that holds the
# SIMD for a whole 4-cycle pass (4.2); "trans": transcendentals and v_permlane*_swap (8.2).
SIMPLE = ("v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_add_u32",
          "v_sub_u32", "v_subrev_u32", "v_xor_b32")
TRANS = ("v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_log", "v_sin", "v_cos", "v_permlane")


def issue_class(ins):
    op = ins.split()[0]
    if not op.startswith("v_"):
        return None
    if op.startswith(TRANS):
        return "trans"
    if "_dpp" in ins or "row_" in ins or "quad_perm" in ins:
        return "dpp"
    base = op.replace("_e32", "").replace("_e64", "")
    if base in SIMPLE and "_sdwa" not in ins:
        return "simple"
    return "full"


# loops: a branch in block j to a label at block i <= j
loops = []
for j, (n, ins) in enumerate(blocks):
    for t in ins:
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in index and index[m.group(1)] <= j:
            loops.append((index[m.group(1)], j))
# innermost loops first: a loop is reported if it does not contain a loop that was reported already
seen = []
report = []
for a, b in sorted(set(loops), key=lambda ab: ab[1] - ab[0]):
    if any(a <= x and y <= b for x, y in seen):
        kind = "outer"
    else:
        kind = "inner"
    c = collections.Counter()
    for _, ins in blocks[a:b + 1]:
        for t in ins:
            c[klass(t)] += 1
    total = sum(c.values())
    valu = sum(v for k, v in c.items() if k.startswith("valu"))
    if total < 60:
        continue
    seen.append((a, b))
    ic = collections.Counter(issue_class(t) for _, ins in blocks[a:b + 1] for t in ins)
    ic.pop(None, None)
    print(f"{kind} loop {blocks[a][0]}..{blocks[b][0]}: {total} instructions, {valu} VALU  " +
          " ".join(f"{k}={v}" for k, v in sorted(c.items())) + "  | issue classes: " +
          " ".join(f"{k}={v}" for k, v in sorted(ic.items())))
    report.append(dict(kind=kind, first=blocks[a][0], last=blocks[b][0], instructions=total, valu=valu,
                       classes=dict(sorted(c.items())), issue_classes=dict(sorted(ic.items()))))
if json_out:
    import json
    import os
    data = json.load(open(json_out)) if os.path.exists(json_out) else {}
    # the hot loop of a kernel = its largest VALU count among the loops that contain no other reported loop ... the caller
    # names it: by convention the FIRST "outer" loop with an atomic (compositing backward) or the first reported loop
    data[json_key] = report
    json.dump(data, open(json_out, "w"), indent=1, sort_keys=True)
