#!/bin/bash
# round 4: the camera launch with drawn chunks (PT_DRAW_CAMERA=1) against dealt ones, resident paths on
set -o pipefail
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout -k 10 900 env PT_DRAW_CAMERA=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident or config1 or geometry_paths or sequences" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  cold {d['value_cold'] or 0:9.0f}  frac {d['roofline']['frac']:.3f}  alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config'].get('resident_paths')} seq {d['config']['launch_sequences_in_flight']}")
PY
}
for rep in 1 2 3; do
b c2_res_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --resident 1 &&
b c2_res_dc_$rep PT_REFILL_MIN=8 PT_DRAW_CAMERA=1 python bench.py --no-cpu-baseline --resident 1 &&
b c2_res_s1_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --resident 1 --sequences 1 &&
b c2_res_dc_s1_$rep PT_REFILL_MIN=8 PT_DRAW_CAMERA=1 python bench.py --no-cpu-baseline --resident 1 --sequences 1 &&
b c5_res_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --resident 1 --config 5 --steps 512 &&
b c5_res_dc_$rep PT_REFILL_MIN=8 PT_DRAW_CAMERA=1 python bench.py --no-cpu-baseline --resident 1 --config 5 --steps 512 &&
b drv_res_$rep PT_REFILL_MIN=8 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 1 &&
b drv_res_dc_$rep PT_REFILL_MIN=8 PT_DRAW_CAMERA=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident 1 || exit 1
done
