mkdir -p gpurun_out/r02ai
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02ai/pytest.log 2>&1; tail -2 gpurun_out/r02ai/pytest.log | cut -c1-200
for c in 2 3; do
python bench.py --config $c --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('config', $c, round(json.loads(l)['value']))
"
done
