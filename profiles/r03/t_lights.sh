O=gpurun_out/${1:-r03e}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
python tests/fuzz_gpu.py 1500 0 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
