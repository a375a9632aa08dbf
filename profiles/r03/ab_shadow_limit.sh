# A/B of the shadow rays' distance limit in the pair pre-test (baseline = same sources with -DPT_NO_SHADOW_LIMIT): GPU suite + 6000 fuzz cases on the product
# build first, then alternating bench runs with direct lighting.  $1 = tag
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 400 python tests/fuzz_gpu.py 6000 1300000 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
bash profiles/r03/ab_multi_cfg.sh $1 "lib_nolimit lib" "--direct-light --steps 128|--direct-light --config 3 --steps 128|--steps 128"
