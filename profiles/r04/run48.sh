#!/bin/bash
# round 4: pre-test's max with the constant 0 instead of the entry's w (one canonicalising v_max less per primitive and trip): A/B against lib_prev3
OUT=gpurun_out/r04zv; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/project3-pathtracer_amd
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2 3; do
for cfgargs in "c2:" "drv:--steps 20 --warmup 5" "c3:--config 3 --steps 256" "c2deg:--rotat degrees"; do
n=${cfgargs%%:*}; a=${cfgargs#*:}
b ${n}_prev_$rep PT_LIBPTAMD=$P/lib_prev3/libptamd.so python bench.py --no-cpu-baseline $a
b ${n}_new_$rep PT_X=0 python bench.py --no-cpu-baseline $a
done; done
