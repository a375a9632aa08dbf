#!/bin/bash
# round 4: resident paths against launch per bounce on small frames (few chunks per wave: the drain weighs more)
OUT=gpurun_out/r04p; mkdir -p $OUT
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} grid {d['config']['grid']} resident {d['config']['resident_paths']}")
PY
}
for rep in 1 2; do
for res in -1 1; do
b c1_r${res}_$rep PT_X=0 python bench.py --no-cpu-baseline --config 1 --steps 256 --resident $res
b w640_r${res}_$rep PT_X=0 python bench.py --no-cpu-baseline --width 640 --height 360 --steps 256 --resident $res
b w960_r${res}_$rep PT_X=0 python bench.py --no-cpu-baseline --width 960 --height 540 --steps 256 --resident $res
b w1280_r${res}_$rep PT_X=0 python bench.py --no-cpu-baseline --width 1280 --height 720 --steps 256 --resident $res
b c1_d8_r${res}_$rep PT_X=0 python bench.py --no-cpu-baseline --config 1 --depth 8 --steps 256 --resident $res
done; done
