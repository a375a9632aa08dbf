# compiler scheduling strategies for the two hot translation units (pair queue, batched walk): parity on each variant (a quick subset), then alternating bench runs
O=gpurun_out/$1; mkdir -p $O
for l in lib_ilp lib_bias0 lib_wprio; do
  PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/$l/libptamd.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config1 or cloud or geometry_paths or random_scenes or fuzz_sample" > $O/pytest_$l.log 2>&1; echo "$l pytest rc=$?"; tail -1 $O/pytest_$l.log
done
bash profiles/r03/ab_multi_cfg.sh $1 "lib lib_ilp lib_bias0 lib_wprio" "--steps 128|--config 5 --steps 128"
