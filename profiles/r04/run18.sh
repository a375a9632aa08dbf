#!/bin/bash
# round 4: the pre-test's box planes picked by ADDRESS (octant table in LDS, product build) against min / max per axis (lib_oct0: -DPT_PRETEST_OCT=0);
# then resident paths against launch per bounce by depth
set -o pipefail
OUT=gpurun_out/r04r; mkdir -p $OUT
OLD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_oct0/libptamd.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not fuzz and not large and not multi_device_gather" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config']['resident_paths']}")
PY
}
for rep in 1 2 3; do
b c2_minmax_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline &&
b c2_oct_$rep PT_X=0 python bench.py --no-cpu-baseline &&
b c3_minmax_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 3 --steps 256 &&
b c3_oct_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256 &&
b drv_minmax_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --steps 20 --warmup 5 &&
b drv_oct_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 || exit 1
done
b dl_minmax PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --direct-light --steps 128
b dl_oct PT_X=0 python bench.py --no-cpu-baseline --direct-light --steps 128
for d in 3 4 5 6; do for res in -1 1; do
b hd_d${d}_r${res} PT_X=0 python bench.py --no-cpu-baseline --depth $d --steps 256 --resident $res
b c1_d${d}_r${res} PT_X=0 python bench.py --no-cpu-baseline --config 1 --depth $d --steps 256 --resident $res
done; done
