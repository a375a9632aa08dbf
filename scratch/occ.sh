for m in 1 2; do
  PT_MAX_WG_PER_CU=$m python bench.py --config 5 --steps 128 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('maxwg', $m, round(json.loads(l)['value']))
"
done
python bench.py --config 5 --steps 128 --workgroup 256 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('wg256', round(json.loads(l)['value']))
"
