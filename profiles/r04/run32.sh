#!/bin/bash
# round 4: carry_max 8 as the default, extreme scenes as a test and a fuzz dimension: suites, fuzz on both builds
set -o pipefail
OUT=gpurun_out/r04zf; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 400 python tests/fuzz_gpu.py 15000 3100000 > $OUT/fuzz_bounds.log 2>&1; tail -1 $OUT/fuzz_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_bounds.log)"; grep -m3 "MISMATCH\|BOUNDS" $OUT/fuzz_bounds.log
timeout -k 10 500 python tests/fuzz_gpu.py 35000 3200000 > $OUT/fuzz.log 2>&1; tail -1 $OUT/fuzz.log; grep -m5 "MISMATCH" $OUT/fuzz.log
