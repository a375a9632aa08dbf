#!/bin/bash
# round 4: refill threshold 4 as the default: suite, then 4 against 8 on every config, one box
set -o pipefail
OUT=gpurun_out/r04zr; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }; tail -1 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2; do
for cfgargs in "c2:" "drv:--steps 20 --warmup 5" "c3:--config 3 --steps 256" "c4:--config 4 --steps 64" "c5:--config 5 --steps 256" "c2deg:--rotat degrees"; do
n=${cfgargs%%:*}; a=${cfgargs#*:}
b ${n}_r8_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline $a
b ${n}_r4_$rep PT_X=0 python bench.py --no-cpu-baseline $a
done; done
