#!/bin/bash
# round 4: carry-over of a trip's leftover pairs (PairCarry, KParams::carry_max): suite on the bounds-checking and the product build, sweep of carry_max
set -o pipefail
OUT=gpurun_out/r04ze; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config']['resident_paths']}")
PY
}
for rep in 1 2; do
for cm in 0 4 8 16 24 32 48; do
b c2_carry${cm}_$rep PT_CARRY_MAX=$cm python bench.py --no-cpu-baseline
done
for cm in 0 16 32; do
b drv_carry${cm}_$rep PT_CARRY_MAX=$cm python bench.py --no-cpu-baseline --steps 20 --warmup 5
b c3_carry${cm}_$rep PT_CARRY_MAX=$cm python bench.py --no-cpu-baseline --config 3 --steps 256
b c2deg_carry${cm}_$rep PT_CARRY_MAX=$cm python bench.py --no-cpu-baseline --rotat degrees
done
done
