"""Instruction mix of a kernel's loop bodies from hipcc's --save-temps assembly:
    python tools/isa_mix.py <file.s> <substring of the mangled kernel name>
Splits the function into basic blocks at labels / branches, keeps the blocks that sit inside a loop (a backward branch
targets them or a later block branches back over them) and prints, per loop body, the instruction classes."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
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


# loops: a branch in block j to a label at block i <= j
loops = []
for j, (n, ins) in enumerate(blocks):
    for t in ins:
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", t)
        if m and m.group(1) in index and index[m.group(1)] <= j:
            loops.append((index[m.group(1)], j))
# innermost loops first: a loop is reported if it does not contain a loop that was reported already
seen = []
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
    print(f"{kind} loop {blocks[a][0]}..{blocks[b][0]}: {total} instructions, {valu} VALU  " +
          " ".join(f"{k}={v}" for k, v in sorted(c.items())))
