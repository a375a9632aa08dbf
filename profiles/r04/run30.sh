#!/bin/bash
# round 4: slab pre-test for tilted cubes (KParams::slab_mask): bounds-checking build on the suite, pair statistics, A/B against PT_NO_SLAB=1
set -o pipefail
OUT=gpurun_out/r04zd; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"
L=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_pair/libptamd.so
for env in "PT_NO_SLAB=1" "PT_X=0"; do
for args in "--depth 8" "--depth 1" "--config 3"; do
  echo "== $env $args"
  env $env PT_LIBPTAMD=$L PT_DEBUG_PAIR=1 timeout -k 10 300 python bench.py --no-cpu-baseline --sequences 1 --steps 32 --warmup 16 --settle-ms 0 $args 2>&1 >/dev/null | grep -E "pair queue" | tail -1
done; done 2>&1 | tee $OUT/pair_stats_slab.txt
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config']['resident_paths']}")
PY
}
for rep in 1 2; do
b c2_noslab_$rep PT_NO_SLAB=1 python bench.py --no-cpu-baseline
b c2_slab_$rep PT_X=0 python bench.py --no-cpu-baseline
b drv_noslab_$rep PT_NO_SLAB=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b drv_slab_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5
b c3_noslab_$rep PT_NO_SLAB=1 python bench.py --no-cpu-baseline --config 3 --steps 256
b c3_slab_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256
b c2deg_noslab_$rep PT_NO_SLAB=1 python bench.py --no-cpu-baseline --rotat degrees
b c2deg_slab_$rep PT_X=0 python bench.py --no-cpu-baseline --rotat degrees
done
