# Round-4 FINAL evidence on ONE box (after self-skip, slab pre-test, carry-over, noise-proof bounds): suites, bench lines of every BASELINE config + variants,
# rocprofv3 kernel traces, PMC passes, fuzz on both builds.   gpurun -- 'bash profiles/r04/evidence2.sh r04fin2'
T=${1:-r04fin2}
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/$T/pytest.log 2>&1; tail -1 gpurun_out/$T/pytest.log
PT_LIBPTAMD=$R/project3-pathtracer_amd/lib_dbg/libptamd.so PT_DEBUG_BOUNDS=1 timeout -k 10 900 python -m pytest tests -q -m gpu -k "not test_abi and not multi_device_gather and not headless" > gpurun_out/$T/pytest_bounds.log 2>&1; tail -1 gpurun_out/$T/pytest_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' gpurun_out/$T/pytest_bounds.log)"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.log 2>&1; tail -2 gpurun_out/$T/smoke.log
bash profiles/r04/evidence.sh $T
PT_LIBPTAMD=$R/project3-pathtracer_amd/lib_dbg/libptamd.so PT_DEBUG_BOUNDS=1 timeout -k 10 400 python tests/fuzz_gpu.py 20000 3400000 > gpurun_out/$T/fuzz_bounds.log 2>&1; tail -1 gpurun_out/$T/fuzz_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' gpurun_out/$T/fuzz_bounds.log)"
timeout -k 10 500 python tests/fuzz_gpu.py 40000 3500000 > gpurun_out/$T/fuzz.log 2>&1; tail -1 gpurun_out/$T/fuzz.log
