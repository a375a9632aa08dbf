for r in 1 2; do
for v in 0 1; do
if [ $v = 1 ]; then export PT_WG896=1; else unset PT_WG896; fi
PT_DEBUG_CLOCK=1 python bench.py --config 5 --steps 256 --no-cpu-baseline 2> gpurun_out/err.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('wg896=$v', round(json.loads(l)['value']))
"
grep -m1 "ptamd" gpurun_out/err.txt | cut -c1-150
done
done
