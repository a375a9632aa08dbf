O=gpurun_out/${1:-r03f}
N=${2:-3000}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
python tests/fuzz_gpu.py $N 0 > $O/fuzz.log 2>&1; grep MISMATCH $O/fuzz.log | head -10; tail -2 $O/fuzz.log
