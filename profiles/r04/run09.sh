#!/bin/bash
# round 4: the batched walk with fp16 node planes (product) against fp32 ones (lib_w4f32: -DPT_W4_HALF=0): parity, rates, LDS counters
set -o pipefail
OUT=gpurun_out/r04i; mkdir -p $OUT
F32=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_w4f32/libptamd.so
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "cloud or walk or mesh or triangle or config5 or resident or geometry_path or random_scenes or fuzz" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f}  alone {d['roofline']['kernel_alone']['frac']:.3f} resident {d['config'].get('resident_paths')} geom {d['config']['geom_path']} wg {d['config']['workgroup']} grid {d['config']['grid']}")
PY
}
for rep in 1 2 3; do
b c5_f32_$rep PT_LIBPTAMD=$F32 python bench.py --no-cpu-baseline --config 5 --steps 512 &&
b c5_f16_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 512 || exit 1
done
b c5_f32_off PT_LIBPTAMD=$F32 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident -1 &&
b c5_f16_off PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 512 --resident -1 &&
b c5_f16_wg256 PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 512 --workgroup 256 &&
b c5_g8 PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 512 --geom-path 8
PT_LIBPTAMD=$F32 python3 profiles/collect_pmc.py --tag r04_c5_f32 --passes sq1,sq2,sq3 -- --config 5 > $OUT/pmc_f32.log 2>&1
python3 profiles/collect_pmc.py --tag r04_c5_f16 --passes sq1,sq2,sq3 -- --config 5 > $OUT/pmc_f16.log 2>&1
for t in f32 f16; do echo "== $t"; python3 - <<PY
import json
s=json.load(open("gpurun_out/pmc_r04_c5_$t/summary.json"))
d=s["k_bounce_derived"]; print({k:(round(v,3) if isinstance(v,float) and v<1000 else v) for k,v in d.items()})
PY
done
