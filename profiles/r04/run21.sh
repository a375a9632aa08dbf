#!/bin/bash
# round 4: knobs again on the build with the octant pre-test: refill threshold, sequences, workgroups per CU
OUT=gpurun_out/r04u; mkdir -p $OUT
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2; do
for k in 2 4 8 12 16; do b c2_k${k}_$rep PT_REFILL_MIN=$k python bench.py --no-cpu-baseline; done
b c2_s3_$rep PT_X=0 python bench.py --no-cpu-baseline --sequences 3
b c2_wg7_$rep PT_MAX_WG_PER_CU=7 python bench.py --no-cpu-baseline
b c2_wg5_$rep PT_MAX_WG_PER_CU=5 python bench.py --no-cpu-baseline
b c2_off_$rep PT_X=0 python bench.py --no-cpu-baseline --resident -1
b drv_k4_$rep PT_REFILL_MIN=4 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b drv_k8_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b drv_off_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident -1
done
