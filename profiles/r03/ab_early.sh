# A/B of the early slot reservation (-DPT_EARLY_RESERVE: the wave's reservation atomic issued right after the nearest hit): parity tests on the
# variant first, then alternating bench runs.  $1 = tag
O=gpurun_out/$1; mkdir -p $O
PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_early/libptamd.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -m gpu -x -q > $O/pytest_early.log 2>&1; echo "pytest(early) rc=$?"; tail -2 $O/pytest_early.log
bash profiles/r03/ab_multi_cfg.sh $1 "lib lib_early" "--steps 128|--config 5 --steps 128|--config 3 --steps 128"
