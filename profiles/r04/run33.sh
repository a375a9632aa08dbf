#!/bin/bash
# round 4: host frustum cull with the fp32 ray-grid margin: the flagged case, the suites, a long fuzz on the bounds-checking build
set -o pipefail
OUT=gpurun_out/r04zg; mkdir -p $OUT
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 300 python -u profiles/r04/dbg_case.py 3103489 3103489 > $OUT/case.log 2>&1; grep -c "BOUNDS" $OUT/case.log; tail -2 $OUT/case.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather" > $OUT/pytest_bounds.log 2>&1 || { tail -40 $OUT/pytest_bounds.log; exit 1; }
tail -1 $OUT/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/pytest_bounds.log)"
timeout -k 10 900 python -m pytest tests -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 700 python tests/fuzz_gpu.py 40000 3300000 > $OUT/fuzz_bounds.log 2>&1; tail -1 $OUT/fuzz_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_bounds.log)"; grep -m3 "MISMATCH\|BOUNDS" $OUT/fuzz_bounds.log
