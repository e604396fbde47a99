"""Print the figures of a bench.py line that matter when comparing runs (tools/*.sh)."""
import json
import sys

try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d["config"]
    k = {n: v["avg_ms"] for n, v in d["kernels_ms"].items()}
    print(f"{sys.argv[2]:28s} {d['value']:9.2f} it/s {d['ms_per_step']:.4f} ms p50 {d['gpu_step_ms']['p50']:.4f} | N {c['n_gaussians']} vis {c['n_visible']} "
          f"M {c['n_isects']} live {c['n_isects_live']} binned/rect {c.get('binned_over_rect_pairs')} L {c.get('mean_walked_list')} "
          f"(max {c.get('max_walked_list')}) never_upd {d.get('never_updated_gaussians_frac')} cuts {d.get('occlusion_cut')} "
          f"hbm_frac {d.get('iter_hbm_frac')} fwd {d.get('fwd_ms')} ms")
    print("    " + " ".join(f"{n.replace('raster_', 'r_').replace('_quad_d4e3', '')}={v:.4f}" for n, v in sorted(k.items())))
    r = d.get("roofline") or {}
    print(f"    roofline: {r.get('kernel', '')[:60]} frac {r.get('frac')} traffic_frac {r.get('traffic_frac_of_peak')} limiter {r.get('limiter')}")
    if d.get("dropin_iters_per_s"):
        print(f"    dropin {d['dropin_iters_per_s']} patched_full {d.get('patched_full_iters_per_s')} detail {d.get('dropin_detail')}")
except Exception as e:  # noqa: BLE001
    print(sys.argv[2], "FAILED", repr(e))
    try:
        print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
    except OSError:
        pass
