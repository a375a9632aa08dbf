#!/bin/bash
# round 4, closing: one more long fuzz on the final kernels (both builds; the generators alone as well)
OUT=gpurun_out/r04zs; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
timeout -k 10 570 python tests/fuzz_gpu.py 70000 4300000 > $OUT/fuzz_a.log 2>&1; tail -1 $OUT/fuzz_a.log; grep -m5 MISMATCH $OUT/fuzz_a.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 400 python tests/fuzz_gpu.py 25000 4400000 > $OUT/fuzz_bounds.log 2>&1; tail -1 $OUT/fuzz_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_bounds.log)"; grep -m5 "MISMATCH\|BOUNDS" $OUT/fuzz_bounds.log
PT_FUZZ_SCENES=stress PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 200 python tests/fuzz_gpu.py 10000 4500000 > $OUT/fuzz_stress_bounds.log 2>&1; tail -1 $OUT/fuzz_stress_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_stress_bounds.log)"
PT_FUZZ_SCENES=extreme PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 200 python tests/fuzz_gpu.py 10000 4600000 > $OUT/fuzz_extreme_bounds.log 2>&1; tail -1 $OUT/fuzz_extreme_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_extreme_bounds.log)"
exit 0
