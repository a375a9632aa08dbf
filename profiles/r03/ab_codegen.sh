# code-generation switches: the whole library built for gfx950:xnack- (the pool runs with XNACK off), units 4 and 6 at -O2 / -Os; parity subset on each, alternating bench runs
O=gpurun_out/$1; mkdir -p $O
for l in lib_xnackoff lib_o2 lib_os; do
  PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/$l/libptamd.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config1 or cloud or geometry_paths or random_scenes or fuzz_sample or glass or specular" > $O/pytest_$l.log 2>&1; echo "$l pytest rc=$?"; tail -1 $O/pytest_$l.log
done
bash profiles/r03/ab_multi_cfg.sh $1 "lib lib_xnackoff lib_o2 lib_os" "--steps 128|--config 5 --steps 128"
