#!/bin/bash
# round 4: pair-queue statistics with resident paths, with and without the self-skip (diagnostic build -DPT_DEBUG_PAIR=1)
OUT=gpurun_out/r04zc; mkdir -p $OUT
L=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_pair/libptamd.so
for env in "PT_NO_SELF_SKIP=1" "PT_X=0"; do
for args in "--depth 8" "--depth 8 --rotat degrees" "--config 3"; do
  echo "== $env $args"
  env $env PT_LIBPTAMD=$L PT_DEBUG_PAIR=1 timeout -k 10 300 python bench.py --no-cpu-baseline --sequences 1 --steps 32 --warmup 16 --settle-ms 0 $args 2>&1 >/dev/null | grep -E "pair queue" | tail -1
done; done 2>&1 | tee $OUT/pair_stats_skip.txt
