#!/bin/bash
# round 4: pair-queue statistics of the camera rays alone (depth 1) and of all bounces (diagnostic build -DPT_DEBUG_PAIR=1 of unit 4)
OUT=gpurun_out/r04n; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_pair/libptamd.so
for args in "--depth 1" "--depth 8 --resident -1" "--depth 1 --rotat degrees" "--depth 8 --resident -1 --rotat degrees" "--config 3 --depth 1" "--config 3 --resident -1"; do
  echo "== $args"
  PT_LIBPTAMD=$L PT_DEBUG_PAIR=1 timeout -k 10 300 python bench.py --no-cpu-baseline --sequences 1 --steps 32 --warmup 16 --settle-ms 0 $args 2>&1 >/dev/null | grep -E "pair queue" | tail -1
done 2>&1 | tee $OUT/pair_stats.txt
