#!/bin/bash
# round 4: carry-over out, slab pre-test as kernel instances of their own (FEAT_SLAB): suites, then old / skip+slab-no-carry (lib_abl_NOCARRY) / new on one box
set -o pipefail
OUT=gpurun_out/r04zn; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/project3-pathtracer_amd
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }; tail -1 $OUT/pytest.log
PT_LIBPTAMD=$P/lib_dbg/libptamd.so PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather and not headless" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2; do
for cfgargs in "c3:--config 3 --steps 256" "c2:" "drv:--steps 20 --warmup 5" "c1:--config 1 --steps 256" "c4:--config 4 --steps 64"; do
n=${cfgargs%%:*}; a=${cfgargs#*:}
b ${n}_old_$rep PT_LIBPTAMD=$P/lib_old/libptamd.so python bench.py --no-cpu-baseline $a
b ${n}_nocarry_$rep PT_LIBPTAMD=$P/lib_abl_NOCARRY/libptamd.so python bench.py --no-cpu-baseline $a
b ${n}_new_$rep PT_X=0 python bench.py --no-cpu-baseline $a
done
b c2_new_noslab_$rep PT_NO_SLAB=1 python bench.py --no-cpu-baseline
b c2_new_noskip_$rep PT_NO_SELF_SKIP=1 python bench.py --no-cpu-baseline
done
