#!/bin/bash
# round 4: self-skip (Prim::self_r2) -- the stress scenes and a long fuzz on the bounds-checking build (every skipped pair's exact test runs; a hit is reported),
# then the same on the product build
set -o pipefail
OUT=gpurun_out/r04za; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 600 python -m pytest tests -q -m gpu -k "self_skip or resident or cloud or random_scenes" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"
timeout -k 10 600 python -m pytest tests -q -m gpu -k "self_skip" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 500 python tests/fuzz_gpu.py 30000 2700000 > $OUT/fuzz_bounds.log 2>&1; tail -1 $OUT/fuzz_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_bounds.log)"
timeout -k 10 500 python tests/fuzz_gpu.py 40000 2800000 > $OUT/fuzz.log 2>&1; tail -1 $OUT/fuzz.log
