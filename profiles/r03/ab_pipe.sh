# A/B of two chunks in flight per wave on the pair path (product, PT_PIPE=1) against the same sources with -DPT_PIPE=0: parity first (GPU suite, then 6000 fuzz
# cases), then alternating bench runs.  $1 = tag
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 400 python tests/fuzz_gpu.py 6000 1900000 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
PT_DEBUG_CLOCK=1 python bench.py --no-cpu-baseline --steps 16 2>&1 >/dev/null | grep "launch:" | sort | uniq
bash profiles/r03/ab_multi_cfg.sh $1 "lib_nopipe lib" "--steps 128|--steps 20 --warmup 5|--config 3 --steps 128|--config 4 --steps 64|--rotat degrees --steps 128"
