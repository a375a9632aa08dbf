#!/bin/bash
# round 4: reach-based pads for triangles (tiny mesh in a huge scene): the three flagged cases, then fuzz runs made of extreme / stress scenes only, both builds
set -o pipefail
OUT=gpurun_out/r04zj; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
for c in 3514219 3505609 3403963; do timeout -k 10 200 python -u tests/fuzz_gpu.py 1 $c 2>&1 | grep -E "MISMATCH|^fuzz"; done
PT_FUZZ_SCENES=extreme timeout -k 10 400 python tests/fuzz_gpu.py 12000 3600000 > $OUT/fuzz_extreme.log 2>&1; tail -1 $OUT/fuzz_extreme.log; grep -m5 "MISMATCH" $OUT/fuzz_extreme.log
PT_FUZZ_SCENES=extreme PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 400 python tests/fuzz_gpu.py 8000 3700000 > $OUT/fuzz_extreme_bounds.log 2>&1; tail -1 $OUT/fuzz_extreme_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_extreme_bounds.log)"; grep -m5 "MISMATCH\|BOUNDS" $OUT/fuzz_extreme_bounds.log
PT_FUZZ_SCENES=stress timeout -k 10 400 python tests/fuzz_gpu.py 12000 3800000 > $OUT/fuzz_stress.log 2>&1; tail -1 $OUT/fuzz_stress.log; grep -m5 "MISMATCH" $OUT/fuzz_stress.log
PT_FUZZ_SCENES=stress PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 400 python tests/fuzz_gpu.py 8000 3900000 > $OUT/fuzz_stress_bounds.log 2>&1; tail -1 $OUT/fuzz_stress_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_stress_bounds.log)"; grep -m5 "MISMATCH\|BOUNDS" $OUT/fuzz_stress_bounds.log
exit 0
