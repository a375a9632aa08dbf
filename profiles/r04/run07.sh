#!/bin/bash
# round 4: full GPU suite on the build with resident paths as the library default; then grids of the resident launch
set -o pipefail
OUT=gpurun_out/r04g; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
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
b c2_res_$rep PT_X=0 python bench.py --no-cpu-baseline &&
for g in 3 4 5; do b c2_res_g${g}_$rep PT_RES_WG_PER_CU=$g python bench.py --no-cpu-baseline || exit 1; done
b c2_res_g4_s3_$rep PT_RES_WG_PER_CU=4 python bench.py --no-cpu-baseline --sequences 3 &&
b c2_res_wg7_$rep PT_MAX_WG_PER_CU=7 python bench.py --no-cpu-baseline &&
b drv_off_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 --resident -1 &&
b drv_res_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 || exit 1
done
