#!/usr/bin/env python3
"""profiles/r6_isa_mix.json: the static vector-instruction mix of the hot loop of each kernel bench.py prices against the
vector-issue roofline (issue classes as measured by tools/valu_rate.hip: simple / full / dpp / trans).

    python tools/isa_mix_all.py            (cross-compiles csrc/*.hip to assembly with the Makefile's flags; no GPU needed)

Hot loop per kernel: the compositing backward's row walk (the first loop body that issues the gradient atomic), else the
innermost loop with the most vector instructions."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fusionsense_amd", "csrc")
NOSLP = {"raster_live", "sh"}
KERNELS = [  # (name in the JSON, source, substring of the mangled name, rule)
    ("raster_bwd_live<4,true,3>", "raster_live", "raster_bwd_live_kernelILi4ELb1ELi3E", "atomic"),
    ("raster_fwd_wave<4,3>", "raster_quad", "raster_fwd_wave_kernelILi4ELi3E", "whole"),
    ("gauss_sh_bwd", "sh", "gauss_sh_bwd_kernelILb0ELb1E", "whole"),
    ("ssim_l1_fwd", "loss", "ssim_l1_fwd_kernel", "whole"),
    ("ssim_l1_bwd", "loss", "ssim_l1_bwd_kernel", "whole"),
    ("isect_live_bin_fill", "isect", "isect_live_bin_fill_kernel", "whole"),
    ("tile_sort_kernel2", "tilesort", "tile_sort_kernel2", "whole"),
]


def main():
    out = {}
    tmp = tempfile.mkdtemp()
    asm = {}
    for name, src, key, rule in KERNELS:
        if src not in asm:
            path = os.path.join(tmp, src + ".s")
            flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-S", "--cuda-device-only"]
            if src in NOSLP:
                flags.append("-fno-slp-vectorize")
            subprocess.run(["/opt/rocm/bin/hipcc"] + flags + [os.path.join(CSRC, src + ".hip"), "-o", path], check=True,
                           stderr=subprocess.DEVNULL)
            asm[src] = path
        j = os.path.join(tmp, "mix.json")
        if os.path.exists(j):
            os.remove(j)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), asm[src], key, "--json", j, "k"],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(j):
            print(f"{name}: not found ({r.stderr.strip().splitlines()[-1] if r.stderr else ''})", file=sys.stderr)
            continue
        rep = json.load(open(j))["k"]
        pick = None
        if rule == "atomic":
            c = [x for x in rep if x["classes"].get("atomic") == 1 and x["classes"].get("valu_dpp", 0) >= 30]
            pick = min(c, key=lambda x: x["instructions"]) if c else None
        elif rule == "max_inner":
            c = [x for x in rep if x["kind"] == "inner"]
            pick = max(c, key=lambda x: x["valu"]) if c else None
        if pick is None:  # the whole kernel: all reported loops that contain no other (inner), summed
            c = [x for x in rep if x["kind"] == "inner"] or rep
            tot = {}
            for x in c:
                for k, v in x["issue_classes"].items():
                    tot[k] = tot.get(k, 0) + v
            pick = dict(first="all inner loops", last="", issue_classes=tot, valu=sum(tot.values()))
        ic = pick["issue_classes"]
        n = max(sum(ic.values()), 1)
        n_rest = max(n - ic.get("trans", 0), 1)
        out[name] = {"loop": f"{pick['first']}..{pick['last']}", "valu": pick["valu"], "issue_classes": ic,
                     "simple_share": round(ic.get("simple", 0) / n_rest, 4), "dpp_share": round(ic.get("dpp", 0) / n_rest, 4),
                     "trans_share": round(ic.get("trans", 0) / n, 4)}
        print(name, out[name])
    json.dump(out, open(os.path.join(ROOT, "profiles", "r6_isa_mix.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
