# A/B of the batched walk's pass order (product: hierarchy first, then the scene-spanning primitives with the ray's best distance as far bound; baseline
# -DPT_W4_WALLS_FIRST = the old order): GPU suite + 6000 fuzz cases (3000 of them forced onto the walk) on the product build, then alternating bench runs.  $1 = tag
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python tests/fuzz_gpu.py 3000 1500000 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
PT_FUZZ_FORCE_GEOM=7 timeout -k 10 300 python tests/fuzz_gpu.py 3000 1600000 > $O/fuzz_walk.log 2>&1; tail -1 $O/fuzz_walk.log
bash profiles/r03/ab_multi_cfg.sh $1 "lib_wallsfirst lib" "--config 5 --steps 128|--config 5 --steps 128 --sequences 1|--config 5 --steps 128 --direct-light"
