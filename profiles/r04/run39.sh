#!/bin/bash
# round 4: where does config 3 lose 5 % against the first half of the round?  compile-time ablations of skip / slab / carry-over, same box
OUT=gpurun_out/r04zm; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/project3-pathtracer_amd
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2; do
for v in old abl_ALL abl_NOSKIP abl_NOSLAB abl_NOCARRY; do
b c3_${v}_$rep PT_LIBPTAMD=$P/lib_$v/libptamd.so python bench.py --no-cpu-baseline --config 3 --steps 256
done
b c3_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256
for v in old abl_ALL abl_NOSKIP abl_NOSLAB abl_NOCARRY; do
b c2_${v}_$rep PT_LIBPTAMD=$P/lib_$v/libptamd.so python bench.py --no-cpu-baseline
done
b c2_new_$rep PT_X=0 python bench.py --no-cpu-baseline
done
