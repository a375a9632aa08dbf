#!/bin/bash
# round 4: lane budget of the resident launch after self-skip + slab + carry-over; refill threshold and workgroup size retuned
OUT=gpurun_out/r04zh; mkdir -p $OUT
for args in "" "--config 3 --steps 64"; do
echo "== config ${args:-2}, PT_DEBUG_PHASE=2 (resident launch), one sequence"
PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_phase2/libptamd.so PT_DEBUG_PHASE2=1 timeout -k 10 300 python bench.py --no-cpu-baseline --sequences 1 --steps 64 --warmup 16 --settle-ms 0 $args 2>&1 >/dev/null | grep -E "lane budget|^\[ptamd\]   " | tail -9
done 2>&1 | tee $OUT/lane_budget.txt
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  frac {d['roofline']['frac']:.3f} alone {d['roofline']['kernel_alone']['frac']:.3f} wg {d['config'].get('workgroup')} grid {d['config'].get('grid')}")
PY
}
for rep in 1 2; do
for rm in 2 4 8 12 16 24; do
b c2_refill${rm}_$rep PT_REFILL_MIN=$rm python bench.py --no-cpu-baseline
done
b c2_wg512_$rep PT_X=0 python bench.py --no-cpu-baseline --workgroup 512
b c2_wg128_$rep PT_X=0 python bench.py --no-cpu-baseline --workgroup 128
done
