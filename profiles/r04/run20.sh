#!/bin/bash
# round 4: primitive types of the pre-test loop from two bit words (product) against one scalar load per trip of the loop (lib_tm0: -DPT_PRETEST_TMASK=0)
set -o pipefail
OUT=gpurun_out/r04t; mkdir -p $OUT
OLD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_tm0/libptamd.so
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
b c2_load_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline &&
b c2_mask_$rep PT_X=0 python bench.py --no-cpu-baseline &&
b c3_load_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 3 --steps 256 &&
b c3_mask_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256 &&
b drv_load_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --steps 20 --warmup 5 &&
b drv_mask_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 || exit 1
done
b c1_load PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 1 --steps 256
b c1_mask PT_X=0 python bench.py --no-cpu-baseline --config 1 --steps 256
b c4_load PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 4 --steps 64
b c4_mask PT_X=0 python bench.py --no-cpu-baseline --config 4 --steps 64
