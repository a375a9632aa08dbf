# A/B of the flag bytes beside the radiance planes (k_accumulate fetches marked entries only; baseline = the same sources with -DPT_NO_LFLAG):
# GPU suite on the product build first, then alternating bench runs.  $1 = tag
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
bash profiles/r03/ab_multi_cfg.sh $1 "lib_noflag lib" "--steps 128|--steps 20 --warmup 5|--config 5 --steps 128|--config 3 --steps 128|--config 4 --steps 64|--direct-light --steps 128"
