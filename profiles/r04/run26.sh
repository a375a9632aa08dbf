#!/bin/bash
# round 4: resident paths skip the primitive a ray just left on its outside (Prim::self_r2): suite on the bounds-checking build (every skipped pair's exact test runs and
# a hit is reported), suite on the product build, A/B against PT_NO_SELF_SKIP=1
set -o pipefail
OUT=gpurun_out/r04z; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"; grep "BOUNDS violation" $OUT/pytest_bounds.log | head -3
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not multi_device_gather" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 300 python tests/fuzz_gpu.py 3000 2600000 > $OUT/fuzz_bounds.log 2>&1; tail -1 $OUT/fuzz_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_bounds.log)"
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config']['resident_paths']}")
PY
}
for rep in 1 2; do
b c2_noskip_$rep PT_NO_SELF_SKIP=1 python bench.py --no-cpu-baseline
b c2_skip_$rep PT_X=0 python bench.py --no-cpu-baseline
b drv_noskip_$rep PT_NO_SELF_SKIP=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b drv_skip_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b c5_noskip_$rep PT_NO_SELF_SKIP=1 python bench.py --no-cpu-baseline --config 5 --steps 256
b c5_skip_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256
b c3_noskip_$rep PT_NO_SELF_SKIP=1 python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_skip_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256
done
