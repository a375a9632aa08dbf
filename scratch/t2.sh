mkdir -p gpurun_out/r02ag
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02ag/pytest.log 2>&1; tail -2 gpurun_out/r02ag/pytest.log | cut -c1-200
python bench.py --config 5 --steps 256 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('c5', round(json.loads(l)['value']))
"
python bench.py --config 5 --steps 256 --geom-path 8 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('c5 path8', round(json.loads(l)['value']))
"
timeout -k 10 300 python tests/fuzz_gpu.py 3000 400000 > gpurun_out/r02ag/fuzz.log 2>&1; tail -1 gpurun_out/r02ag/fuzz.log
