#!/bin/bash
# round 4: resident paths, drawn chunks with three scalars of wave state; walk kernels held to 80 VGPRs
set -o pipefail
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident" > $OUT/pytest_resident.log 2>&1 || { tail -30 $OUT/pytest_resident.log; exit 1; }
tail -2 $OUT/pytest_resident.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  cold {d['value_cold'] or 0:9.0f}  frac {d['roofline']['frac']:.3f}  alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config'].get('resident_paths')} seq {d['config']['launch_sequences_in_flight']} wg {d['config']['workgroup']} grid {d['config']['grid']}")
PY
}
for rep in 1 2; do
b c5_off_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident -1 &&
b c5_res_k8_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident 1 &&
b c5_res_k1_$rep PT_REFILL_MIN=1 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident 1 &&
b c5_res_k16_$rep PT_REFILL_MIN=16 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident 1 &&
b c5_res_k8_s1_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident 1 --sequences 1 &&
b c2_off_$rep PT_X=0 python bench.py --no-cpu-baseline --resident -1 &&
b c2_res_k8_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --resident 1 &&
b c2_res_k12_$rep PT_REFILL_MIN=12 python bench.py --no-cpu-baseline --resident 1 || exit 1
done
for rep in 1 2 3 4; do
b drv_off_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident -1 &&
b drv_res_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 1 &&
b drv_res_s1_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 1 --sequences 1 &&
b drv_res_b5_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 1 --batch 5 || exit 1
done
