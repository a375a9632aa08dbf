#!/bin/bash
# round 4: resident paths against launch per bounce by depth (1920x1080 and 400x400)
OUT=gpurun_out/r04q; mkdir -p $OUT
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} resident {d['config']['resident_paths']}")
PY
}
for d in 3 4 5 6; do for res in -1 1; do
b hd_d${d}_r${res} PT_X=0 python bench.py --no-cpu-baseline --depth $d --steps 256 --resident $res
b c1_d${d}_r${res} PT_X=0 python bench.py --no-cpu-baseline --config 1 --depth $d --steps 256 --resident $res
done; done
for res in -1 1; do b w1280b_r${res} PT_X=0 python bench.py --no-cpu-baseline --width 1280 --height 720 --steps 256 --resident $res; b w1600_r${res} PT_X=0 python bench.py --no-cpu-baseline --width 1600 --height 900 --steps 256 --resident $res; done
