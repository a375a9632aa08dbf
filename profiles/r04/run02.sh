#!/bin/bash
# round 4: (1) workgroup lifetimes of the bounce-1 launch (the resident launch when on) -- diagnostic build -DPT_DEBUG_SPAN=1;
# (2) the driver's 20-step call cut into batches in different ways
set -o pipefail
OUT=gpurun_out/r04b; mkdir -p $OUT
SPAN=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_span/libptamd.so
for res in -1 1; do for seq in 1 2; do
  echo "== lifetimes: resident $res sequences $seq"
  PT_LIBPTAMD=$SPAN PT_DEBUG_SPAN=1 PT_REFILL_MIN=16 timeout -k 10 300 python bench.py --no-cpu-baseline --resident $res --sequences $seq --steps 64 --warmup 64 --settle-ms 0 2>&1 >/dev/null | grep "lifetimes" | tail -2
done; done
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  cold {d['value_cold'] or 0:9.0f}  frac {d['roofline']['frac']:.3f}  resident {d['config'].get('resident_paths')} batches {d['config']['timed_batches']} seq {d['config']['launch_sequences_in_flight']}")
PY
}
for rep in 1 2; do
for res in -1 1; do for seq in 2 3; do for batch in 0 4 5 7; do
  b drv_r${res}_s${seq}_b${batch}_$rep PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident $res --sequences $seq --batch $batch || exit 1
done; done; done; done
