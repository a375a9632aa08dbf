for m in 6 7 8 6 7; do
PT_MAX_WG_PER_CU=$m python bench.py --config 2 --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('maxwg', $m, round(json.loads(l)['value']))
"
done
