#!/bin/bash
# round 4: the skip's compare behind a wave-uniform "does any lane skip anything" -- config 3 (no self pair to remove) against lib_prev2 (the build before)
OUT=gpurun_out/r04zo; mkdir -p $OUT
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
b c3_old_$rep PT_LIBPTAMD=$P/lib_old/libptamd.so python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_prev_$rep PT_LIBPTAMD=$P/lib_prev2/libptamd.so python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256
b c2_prev_$rep PT_LIBPTAMD=$P/lib_prev2/libptamd.so python bench.py --no-cpu-baseline
b c2_new_$rep PT_X=0 python bench.py --no-cpu-baseline
done
