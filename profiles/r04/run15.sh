#!/bin/bash
# round 4, VERDICT r3 item 4(i): the accumulates on a high-priority stream of their own (PT_ACC_STREAM=1) against in-line on the sequences' streams
set -o pipefail
OUT=gpurun_out/r04o; mkdir -p $OUT
timeout -k 10 600 env PT_ACC_STREAM=1 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sequences or resident or running_mean" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  cold {d['value_cold'] or 0:9.0f}  frac {d['roofline']['frac']:.3f}")
PY
}
for rep in 1 2 3; do
b c2_inline_$rep PT_X=0 python bench.py --no-cpu-baseline &&
b c2_accstream_$rep PT_ACC_STREAM=1 python bench.py --no-cpu-baseline &&
b drv_inline_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 &&
b drv_accstream_$rep PT_ACC_STREAM=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 &&
b c5_inline_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 256 &&
b c5_accstream_$rep PT_ACC_STREAM=1 python bench.py --no-cpu-baseline --config 5 --steps 256 || exit 1
done
