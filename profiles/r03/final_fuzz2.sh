O=gpurun_out/${1:-r03w}
N=${2:-40000}
mkdir -p $O
echo "== bounds-checking build: GPU suite" > $O/fuzz_final.txt
PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so PT_DEBUG_BOUNDS=1 python -m pytest tests -m gpu -x -q > $O/pytest_bounds.log 2>&1; echo "rc=$?" >> $O/fuzz_final.txt; tail -1 $O/pytest_bounds.log >> $O/fuzz_final.txt; grep -c "BOUNDS violation" $O/pytest_bounds.log >> $O/fuzz_final.txt
echo "== bounds-checking build: fuzz 15000 cases from 1700000" >> $O/fuzz_final.txt
PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so PT_DEBUG_BOUNDS=1 python tests/fuzz_gpu.py 15000 1700000 > $O/fuzz_bounds.log 2>&1; tail -1 $O/fuzz_bounds.log >> $O/fuzz_final.txt; grep -c "BOUNDS violation" $O/fuzz_bounds.log >> $O/fuzz_final.txt
echo "== product build: fuzz $N cases from 1800000" >> $O/fuzz_final.txt
python tests/fuzz_gpu.py $N 1800000 > $O/fuzz.log 2>&1; grep MISMATCH $O/fuzz.log | head -5 >> $O/fuzz_final.txt; tail -1 $O/fuzz.log >> $O/fuzz_final.txt
cat $O/fuzz_final.txt
