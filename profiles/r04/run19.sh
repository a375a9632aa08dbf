#!/bin/bash
# round 4: the pre-test loop software-pipelined (next primitive's planes in flight, type from the LDS entry: product build) against the plain octant loop (lib_pipe0)
set -o pipefail
OUT=gpurun_out/r04s; mkdir -p $OUT
OLD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_pipe0/libptamd.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "not fuzz and not large and not multi_device_gather" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config']['resident_paths']}")
PY
}
for rep in 1 2 3; do
b c2_plain_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline &&
b c2_pipe_$rep PT_X=0 python bench.py --no-cpu-baseline &&
b c3_plain_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 3 --steps 256 &&
b c3_pipe_$rep PT_X=0 python bench.py --no-cpu-baseline --config 3 --steps 256 &&
b drv_plain_$rep PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --steps 20 --warmup 5 &&
b drv_pipe_$rep PT_X=0 python bench.py --no-cpu-baseline --steps 20 --warmup 5 || exit 1
done
b c1_plain PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 1 --steps 256
b c1_pipe PT_X=0 python bench.py --no-cpu-baseline --config 1 --steps 256
b c4_plain PT_LIBPTAMD=$OLD python bench.py --no-cpu-baseline --config 4 --steps 64
b c4_pipe PT_X=0 python bench.py --no-cpu-baseline --config 4 --steps 64
