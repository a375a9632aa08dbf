O=gpurun_out/${1:-r03lf4}
mkdir -p $O
( PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so PT_DEBUG_BOUNDS=1 timeout -k 10 1000 python tests/fuzz_gpu.py 40000 2000000 > $O/fuzz_bounds.log 2>&1 ) &
timeout -k 10 1000 python tests/fuzz_gpu.py 100000 2100000 > $O/fuzz_a.log 2>&1 &
timeout -k 10 1000 python tests/fuzz_gpu.py 100000 2200000 > $O/fuzz_b.log 2>&1 &
while [ -n "$(jobs -r)" ]; do sleep 50; tail -q -n1 $O/fuzz_a.log $O/fuzz_b.log $O/fuzz_bounds.log | cut -c1-60; done
wait
echo "== bounds build, 40000 cases from 2000000:" > $O/long_fuzz.txt; tail -1 $O/fuzz_bounds.log >> $O/long_fuzz.txt; grep -c "BOUNDS violation" $O/fuzz_bounds.log >> $O/long_fuzz.txt
echo "== product build, 100000 cases from 2100000:" >> $O/long_fuzz.txt; grep MISMATCH $O/fuzz_a.log | head -3 >> $O/long_fuzz.txt; tail -1 $O/fuzz_a.log >> $O/long_fuzz.txt
echo "== product build, 100000 cases from 2200000:" >> $O/long_fuzz.txt; grep MISMATCH $O/fuzz_b.log | head -3 >> $O/long_fuzz.txt; tail -1 $O/fuzz_b.log >> $O/long_fuzz.txt
cat $O/long_fuzz.txt
