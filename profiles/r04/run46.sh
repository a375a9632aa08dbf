#!/bin/bash
# round 4: config 5 with the walk's nodes through L1/L2 (geom_path 8) instead of the LDS copy (7): does relieving the LDS pipe pay?
OUT=gpurun_out/r04zt; mkdir -p $OUT
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} path {d['config'].get('geom_path')} wg {d['config'].get('workgroup')} grid {d['config'].get('grid')}")
PY
}
for rep in 1 2; do
b c5_g7_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256
b c5_g8_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256 --geom-path 8
b c5_g8_wg256_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256 --geom-path 8 --workgroup 256
b c5_g8_wg512_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256 --geom-path 8 --workgroup 512
b c5_g7_wg256_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256 --workgroup 256
done
