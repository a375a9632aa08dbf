#!/bin/bash
# round 4: workgroup lifetimes of the resident launch on configs 3 and 5 (long paths: how long is the drain?)
OUT=gpurun_out/r04l; mkdir -p $OUT
SPAN=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_span/libptamd.so
for cfg in 5 3; do for seq in 1 2; do
  echo "== config $cfg sequences $seq"
  PT_LIBPTAMD=$SPAN PT_DEBUG_SPAN=1 timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg --sequences $seq --steps 64 --warmup 64 --settle-ms 0 2>&1 >/dev/null | grep "lifetimes" | tail -2
done; done 2>&1 | tee $OUT/lifetimes.txt
python bench.py --no-cpu-baseline > $OUT/c2.json 2> $OUT/c2.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04l/c2.json").read().strip().splitlines()[-1]); print(round(d['value']), json.dumps(d['roofline']['kernel_alone']['per_kernel'], indent=0)[:900])
PY
