#!/bin/bash
# round 4: what do the noise-proof bounds cost?  PT_NO_NOISE_PAD=1 (rounds 1-3's bounds) against the default, same box, + the first-half build
OUT=gpurun_out/r04zl; mkdir -p $OUT
OLD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_old/libptamd.so
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2; do
b c3_old_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_nopad_$rep PT_NO_NOISE_PAD=1 python bench.py --no-cpu-baseline --config 3 --steps 256
b c2_new_$rep PT_X=0 python bench.py --no-cpu-baseline
b c2_nopad_$rep PT_NO_NOISE_PAD=1 python bench.py --no-cpu-baseline
b c5_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256
b c5_nopad_$rep PT_NO_NOISE_PAD=1 python bench.py --no-cpu-baseline --config 5 --steps 256
done
