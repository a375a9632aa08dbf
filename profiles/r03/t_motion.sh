O=gpurun_out/${1:-r03mo}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python tests/fuzz_gpu.py 6000 700000 > $O/fuzz.log 2>&1; grep MISMATCH $O/fuzz.log | head -5; tail -1 $O/fuzz.log
python profiles/r03/motion_rate.py 2>/dev/null
