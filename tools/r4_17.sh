#!/bin/bash
O=gpurun_out/r4q; mkdir -p $O
export TMPDIR=/tmp
for rep in 1 2 3 4; do for l in libfsgs.so libfsgs_nt.so libfsgs_nt3.so; do
  FSGS_LIB=$PWD/fusionsense_amd/$l timeout 900 python bench.py --config 4 --steps 12 --warmup 4 --no-cpu-baseline > $O/c4_$l.$rep.json 2> $O/c4_$l.$rep.err
  python - "$O/c4_$l.$rep.json" "$l" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels_ms']
print(f"{sys.argv[2]:16s} {d['value']:7.2f} it/s  gauss_bwd {k['gaussian_bwd']['avg_ms']:.4f}  r_fwd {k['raster_fwd_quad_d4e3']['avg_ms']:.4f} r_bwd {k['raster_bwd_quad_d4e3']['avg_ms']:.4f} count {k['isect_count_live']['avg_ms']:.4f} sort {k['tile_sort']['avg_ms']:.4f}")
PY
done; done
