#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one or more passes) as CSV, plus the
HBM traffic per launch that bench.py reports as `roofline.traffic`.

Usage: tools/pmc_summary.py OUT_CSV OUT_JSON CONFIG SOURCE_TAG counter_collection.csv [more.csv ...]
(OUT_JSON holds {"config<CONFIG>": {span: {...}}}; merge several into profiles/pmc_traffic.json)

Traffic model (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950
FETCH_SIZE tallies 128-byte requests at 64 B, so it is doubled; WRITE_SIZE is taken as reported."""
import collections
import csv
import json
import sys

# kernel symbol -> bench.py span key (several symbols may share a span)
SPAN = [
    ("gauss_sh_bwd_kernel<false, true>", "gaussian_bwd_adam"), ("gauss_sh_bwd_kernel<true, true>", "gaussian_bwd_adam"),
    ("gauss_sh_bwd_kernel", "gaussian_bwd"),  # (before "sh_bwd_kernel", which is a substring of it)
    ("raster_bwd_live_kernel<4, true, 3>", "raster_bwd_quad_d4e3"),
    ("raster_fwd_quad_kernel<4, 3>", "raster_fwd_quad_d4e3"), ("raster_fwd_wave_kernel<4, 3>", "raster_fwd_quad_d4e3"),
    ("tile_hist_kernel", "tile_sort"), ("tile_scan_kernel2", "tile_sort"), ("tile_offsets_kernel", "tile_sort"),
    ("tile_scatter_kernel", "tile_sort"), ("tile_sort_kernel2", "tile_sort"),
    ("adam_kernel", "adam_step"), ("sh_bwd_kernel", "sh_bwd_split"), ("sh_fwd_kernel", "sh_fwd_split"),
    ("fusion_aux_kernel", "fusion_aux_loss"), ("split_slabs_kernel", "tile_sort"),
    ("tile_split_sort_lds_kernel", "tile_sort"), ("slab_split_sort_lds_kernel", "tile_sort"),
    ("tile_sort_strided_kernel", "tile_sort"), ("split_base_kernel", "tile_sort"),
    ("ssim_l1_fwd_kernel", "ssim_l1_fwd"), ("ssim_l1_bwd_kernel", "ssim_l1_bwd"),
    ("project_bwd_kernel<true>", "gaussian_bwd"),
    ("sh_fwd_pack_direct_kernel", "sh_fwd_split"), ("sh_bwd_hybrid_kernel", "sh_bwd_split"), ("project_fwd_kernel<true>", "project_fwd_act"),
    ("isect_live_flat_kernel<false>", "isect_count_live"), ("isect_live_flat_kernel<true>", "isect_emit_live"),
    ("isect_live_bin_kernel<false", "isect_count_live"), ("isect_live_bin_kernel<true", "tile_sort"),
    ("tile_scan_rows_kernel", "isect_count_live"), ("isect_count_adam_kernel", "isect_count_live"),
    ("scan_rows_sh_pack_kernel", "isect_count_live"),
    ("live_pack_kernel", "live_pack_normals_d4e3"),
]


ALTERNATIVES = ("raster_fwd_quad_d4e3",)


def main():
    out_csv, out_json, config, source, files = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for d in acc.values() for c in d})
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches"] + [c + "_avg" for c in counters])
        for k, d in sorted(acc.items()):
            if "fsgs" not in k:
                continue
            n = max(len(v) for v in d.values())
            w.writerow([k.split("(")[0], n] + [round(sum(d[c]) / len(d[c]), 3) if d[c] else "" for c in counters])
    # symbols that are ALTERNATIVES for one span (the forward compositing's two walks: the trainer alternates them on a
    # few early frames, then keeps one): only the symbol with the most launches describes the span
    for span in ALTERNATIVES:
        syms = [k for k in acc if any(sym in k and sp == span for sym, sp in SPAN)]
        if len(syms) > 1:
            keep = max(syms, key=lambda k: max(len(v) for v in acc[k].values()))
            for k in syms:
                if k != keep:
                    del acc[k]
    traffic = collections.defaultdict(lambda: {"fetch_kb": 0.0, "write_kb": 0.0})
    for k, d in acc.items():
        for sym, span in SPAN:
            if sym in k:
                # launches per step differ between symbols of one span: normalise by the span's dominant launch count
                f = d.get("FETCH_SIZE", [])
                wv = d.get("WRITE_SIZE", [])
                traffic[span]["fetch_kb"] += sum(f)
                traffic[span]["write_kb"] += sum(wv)
                traffic[span].setdefault("n", 0)
                traffic[span]["n"] = max(traffic[span]["n"], len(f), len(wv))
                break
    out = {}
    for span, t in traffic.items():
        if not t.get("n"):
            continue  # this pass did not collect FETCH_SIZE / WRITE_SIZE
        n = max(t.get("n", 1), 1)
        # tile_sort: 6 launches per call, 5 different symbols -> calls = launches of the histogram kernel
        if span == "tile_sort":
            # (one of these runs once per frame: the offsets kernel until round 4, the bucket fill since)
            per_frame = [len(d.get("FETCH_SIZE", d.get("WRITE_SIZE", []))) for k, d in acc.items()
                         if "tile_offsets_kernel" in k or "isect_live_bin_kernel<true" in k or "tile_scatter_kernel" in k]
            n = max(per_frame) if per_frame else n
        fetch_b = 2.0 * t["fetch_kb"] * 1024.0 / n
        write_b = t["write_kb"] * 1024.0 / n
        out[span] = {"fetch_bytes_per_launch": round(fetch_b), "write_bytes_per_launch": round(write_b),
                     "hbm_bytes_per_launch": round(fetch_b + write_b), "launches_sampled": n, "source": source}
    # vector-ALU counters of single-symbol spans (a separate --pmc pass): instructions and busy quad-cycles per launch
    for k, d in acc.items():
        for sym, span in SPAN:
            if sym in k and span != "tile_sort" and d.get("SQ_INSTS_VALU") and span in out:
                out[span]["sq_insts_valu_per_launch"] = round(sum(d["SQ_INSTS_VALU"]) / len(d["SQ_INSTS_VALU"]))
                a = d.get("SQ_ACTIVE_INST_VALU")
                if a:
                    out[span]["sq_active_inst_valu_quadcycles_per_launch"] = round(sum(a) / len(a))
                break
    out["_note"] = ("per-launch averages of rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE, "
                    "SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU), each counter set in its own run of bench.py; made by tools/collect.sh "
                    "+ tools/pmc_summary.py")
    json.dump({f"config{config}": out}, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
