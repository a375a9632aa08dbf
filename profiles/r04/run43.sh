#!/bin/bash
# round 4: refill threshold on configs 3 and 5 (final build)
OUT=gpurun_out/r04zq; mkdir -p $OUT
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2; do
for rm in 2 4 8 12 16 24; do b c3_refill${rm}_$rep PT_REFILL_MIN=$rm python bench.py --no-cpu-baseline --config 3 --steps 256; done
for rm in 2 4 8 16 24; do b c5_refill${rm}_$rep PT_REFILL_MIN=$rm python bench.py --no-cpu-baseline --config 5 --steps 256; done
done
