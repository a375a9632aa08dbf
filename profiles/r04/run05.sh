#!/bin/bash
# round 4: lane budget of the later bounces (diagnostic build -DPT_DEBUG_PHASE=2 of unit 4), launch per bounce and resident paths
OUT=gpurun_out/r04e; mkdir -p $OUT
P2=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_phase2/libptamd.so
for cfg in 2 3; do for res in -1 1; do
  echo "== config $cfg, resident $res, one sequence (diagnostic build: the stamps serialise the wave, rates are not the product's)"
  PT_LIBPTAMD=$P2 PT_DEBUG_PHASE2=1 PT_REFILL_MIN=8 timeout -k 10 300 python bench.py --no-cpu-baseline --config $cfg --resident $res --sequences 1 --steps 64 --warmup 16 --settle-ms 0 2>&1 >/dev/null | grep -E "lane budget|^\[ptamd\]   " | tail -9
done; done 2>&1 | tee $OUT/lane_budget.txt
