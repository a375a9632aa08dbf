#!/bin/bash
# round 4: the ONE closing fuzz run -- product build and bounds-checking build (make OUT=../lib_dbg EXTRA_HIPFLAGS=-DPT_DEBUG_BOUNDS=1), three processes side by side
OUT=gpurun_out/r04fuzz; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
( PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 1000 python tests/fuzz_gpu.py 30000 2000000 > $OUT/fuzz_bounds.log 2>&1; echo "rc=$?" >> $OUT/fuzz_bounds.log ) &
P1=$!
( timeout -k 10 1000 python tests/fuzz_gpu.py 35000 2100000 > $OUT/fuzz_product_a.log 2>&1; echo "rc=$?" >> $OUT/fuzz_product_a.log ) &
P2=$!
( timeout -k 10 1000 python tests/fuzz_gpu.py 35000 2200000 > $OUT/fuzz_product_b.log 2>&1; echo "rc=$?" >> $OUT/fuzz_product_b.log ) &
P3=$!
while kill -0 $P1 2>/dev/null || kill -0 $P2 2>/dev/null || kill -0 $P3 2>/dev/null; do sleep 45; tail -q -n 1 $OUT/fuzz_bounds.log $OUT/fuzz_product_a.log $OUT/fuzz_product_b.log | tr '\n' '|'; echo; done
for f in bounds product_a product_b; do echo "== $f"; grep -E "^fuzz:|MISMATCH|rc=" $OUT/fuzz_$f.log | tail -5; grep -c "BOUNDS violation" $OUT/fuzz_$f.log; done
