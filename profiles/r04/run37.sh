#!/bin/bash
# round 4: first half of the round (commit fa3b9d1, lib_old) against the current build on one box; ablations of the second half's changes on config 3
OUT=gpurun_out/r04zk; mkdir -p $OUT
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
b c3_noskip_$rep PT_NO_SELF_SKIP=1 python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_noslab_$rep PT_NO_SLAB=1 python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_nocarry_$rep PT_CARRY_MAX=0 python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_none_$rep PT_CARRY_MAX=0 PT_NO_SLAB=1 PT_NO_SELF_SKIP=1 python bench.py --no-cpu-baseline --config 3 --steps 256
b c2_old_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline
b c2_new_$rep PT_X=0 python bench.py --no-cpu-baseline
b c4_old_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 4 --steps 64
b c4_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 4 --steps 64
b c5_old_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 5 --steps 256
b c5_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256
b c1_old_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 1 --steps 256
b c1_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 1 --steps 256
done
