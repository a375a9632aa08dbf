mkdir -p gpurun_out/r02af
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02af/pytest.log 2>&1; tail -2 gpurun_out/r02af/pytest.log | cut -c1-200
for wg in 0 512 256; do
PT_DEBUG_CLOCK=1 python bench.py --config 5 --steps 256 --workgroup $wg --no-cpu-baseline 2> gpurun_out/r02af/err_$wg.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('wg', $wg, round(json.loads(l)['value']))
"
grep -m1 "ptamd" gpurun_out/r02af/err_$wg.txt | cut -c1-250
done
