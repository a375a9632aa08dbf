timeout -k 10 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -1 | cut -c1-100
for r in 1 2; do
PT_DEBUG_CLOCK=1 python bench.py --config 5 --steps 256 --no-cpu-baseline 2> gpurun_out/err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('c5', round(json.loads(l)['value']))
"
done
grep -m1 "ptamd" gpurun_out/err.txt | cut -c1-250
