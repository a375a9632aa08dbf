#!/bin/bash
# round 4: the pre-test loop takes a primitive's type and slab bit from a nibble word (one scalar load per eight primitives): suite on the bounds build,
# A/B against the previous build (lib_prev)
set -o pipefail
OUT=gpurun_out/r04zi; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
PREV=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_prev/libptamd.so
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }; tail -1 $OUT/pytest.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather and not headless" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2 3; do
b c2_prev_$rep PT_LIBPTAMD=$PREV python bench.py --no-cpu-baseline
b c2_new_$rep PT_X=0 python bench.py --no-cpu-baseline
b drv_prev_$rep PT_LIBPTAMD=$PREV python bench.py --no-cpu-baseline --steps 20 --warmup 5
b drv_new_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b c3_prev_$rep PT_LIBPTAMD=$PREV python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_new_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256
done
