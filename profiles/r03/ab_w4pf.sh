# A/B of the batched walk with the next head batch fetched a step ahead and the push rank counted onto the ring position (-DPT_W4_PREFETCH, unit 6):
# parity on the variant (whole parity file + bench shapes + 3000 fuzz cases forced onto the walk), then alternating bench runs.  $1 = tag
O=gpurun_out/$1; mkdir -p $O; V=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_w4pf/libptamd.so
PT_LIBPTAMD=$V python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
PT_LIBPTAMD=$V PT_FUZZ_FORCE_GEOM=7 timeout -k 10 300 python tests/fuzz_gpu.py 3000 1400000 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
bash profiles/r03/ab_multi_cfg.sh $1 "lib lib_w4pf" "--config 5 --steps 128|--config 5 --steps 128 --direct-light"
