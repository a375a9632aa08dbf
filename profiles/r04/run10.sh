#!/bin/bash
# round 4: v_fma_mix_f32 issue rate; fp16 node planes read by address (lib_w4h2: -DPT_W4_HALF=2) against the cndmask form (product) and fp32 (lib_w4f32)
OUT=gpurun_out/r04j; mkdir -p $OUT
./profiles/microbench/fma_mix_rate | tee $OUT/fma_mix_rate.txt
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f}  alone {d['roofline']['kernel_alone']['frac']:.3f}")
PY
}
for rep in 1 2 3; do
b c5_f32_$rep PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_w4f32/libptamd.so python bench.py --no-cpu-baseline --config 5 --steps 512 &&
b c5_f16_$rep PT_X=0 python bench.py --no-cpu-baseline --config 5 --steps 512 &&
b c5_h2_$rep PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_w4h2/libptamd.so python bench.py --no-cpu-baseline --config 5 --steps 512 || exit 1
done
