#!/bin/bash
# round 4: lane budget of the camera launch (-DPT_DEBUG_PHASE=3) and, again, of the resident launch (=2)
OUT=gpurun_out/r04m; mkdir -p $OUT
for ph in 3 2; do
  echo "== config 2, PT_DEBUG_PHASE=$ph ($([ $ph = 3 ] && echo camera launch || echo resident launch)), one sequence"
  PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_phase$ph/libptamd.so PT_DEBUG_PHASE2=1 timeout -k 10 300 python bench.py --no-cpu-baseline --sequences 1 --steps 64 --warmup 16 --settle-ms 0 2>&1 >/dev/null | grep -E "lane budget|^\[ptamd\]   " | tail -9
done 2>&1 | tee $OUT/lane_budget2.txt
