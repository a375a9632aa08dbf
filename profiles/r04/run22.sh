#!/bin/bash
# round 4: the octant table's near entry read as 16 bytes (w = the 0 of the slab test: lib_b128) against 12 (product): LDS bank conflicts
OUT=gpurun_out/r04v; mkdir -p $OUT
NEW=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_b128/libptamd.so
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  cold {d['value_cold'] or 0:9.0f} frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} lds {d['config'].get('lds_bytes_per_workgroup')}")
PY
}
PT_LIBPTAMD=$NEW timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config1 or resident or geometry_paths" 2>&1 | tail -1
for rep in 1 2 3; do
b c2_b96_$rep PT_X=0 python bench.py --no-cpu-baseline
b c2_b128_$rep PT_LIBPTAMD=$NEW python bench.py --no-cpu-baseline
b drv_b96_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b drv_b128_$rep PT_LIBPTAMD=$NEW python bench.py --no-cpu-baseline --steps 20 --warmup 5
done
b c5_lds PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 128
