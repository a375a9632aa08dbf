#!/bin/bash
# round 4: second fuzz run, on the last build -- the resident-path instances with direct lighting / scattering were added after the closing run
OUT=gpurun_out/r04fuzz2; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
( PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 700 python tests/fuzz_gpu.py 12000 2300000 > $OUT/fuzz_bounds.log 2>&1; echo "rc=$?" >> $OUT/fuzz_bounds.log ) &
P1=$!
( timeout -k 10 700 python tests/fuzz_gpu.py 14000 2400000 > $OUT/fuzz_product_a.log 2>&1; echo "rc=$?" >> $OUT/fuzz_product_a.log ) &
P2=$!
( timeout -k 10 700 python tests/fuzz_gpu.py 14000 2500000 > $OUT/fuzz_product_b.log 2>&1; echo "rc=$?" >> $OUT/fuzz_product_b.log ) &
P3=$!
while kill -0 $P1 2>/dev/null || kill -0 $P2 2>/dev/null || kill -0 $P3 2>/dev/null; do sleep 45; tail -q -n 1 $OUT/fuzz_bounds.log $OUT/fuzz_product_a.log $OUT/fuzz_product_b.log | tr '\n' '|'; echo; done
for f in bounds product_a product_b; do echo "== $f"; grep -E "^fuzz:|MISMATCH|rc=" $OUT/fuzz_$f.log | tail -5; grep -c "BOUNDS violation" $OUT/fuzz_$f.log; done
