#!/bin/bash
# round 4: resident paths with DRAWN chunks: parity, workgroup lifetimes, rates
set -o pipefail
OUT=gpurun_out/r04c; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident" > $OUT/pytest_resident.log 2>&1 || { tail -30 $OUT/pytest_resident.log; exit 1; }
tail -2 $OUT/pytest_resident.log
SPAN=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_span/libptamd.so
for seq in 1 2; do
  echo "== lifetimes: resident 1 sequences $seq"
  PT_LIBPTAMD=$SPAN PT_DEBUG_SPAN=1 PT_REFILL_MIN=16 timeout -k 10 300 python bench.py --no-cpu-baseline --resident 1 --sequences $seq --steps 64 --warmup 64 --settle-ms 0 2>&1 >/dev/null | grep "lifetimes" | tail -2
done
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  cold {d['value_cold'] or 0:9.0f}  frac {d['roofline']['frac']:.3f}  alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config'].get('resident_paths')} seq {d['config']['launch_sequences_in_flight']}")
PY
}
for rep in 1 2; do
b c2_off_$rep PT_X=0 python bench.py --no-cpu-baseline --resident -1 &&
for k in 1 8 16 24; do b c2_res_k${k}_$rep PT_REFILL_MIN=$k python bench.py --no-cpu-baseline --resident 1 || exit 1; done
b c2_res_seq1_$rep PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --resident 1 --sequences 1 &&
b c2_res_seq3_$rep PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --resident 1 --sequences 3 &&
b drv_off_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident -1 &&
b drv_res_$rep PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 1 &&
b drv_res_s3_$rep PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 1 --sequences 3 || exit 1
done
b c5_off PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident -1 &&
b c5_res_k8 PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident 1 &&
b c5_res_k1 PT_REFILL_MIN=1 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident 1 &&
b c3_off PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256 --resident -1 &&
b c3_res_k16 PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --config 3 --steps 256 --resident 1 &&
b c4_off PT_X=0 python bench.py --no-cpu-baseline --config 4 --steps 64 --resident -1 &&
b c4_res_k16 PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --config 4 --steps 64 --resident 1
