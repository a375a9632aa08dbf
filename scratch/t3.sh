mkdir -p gpurun_out/r02ah
timeout -k 10 300 python -m pytest tests -m gpu -x -q > gpurun_out/r02ah/pytest.log 2>&1; tail -2 gpurun_out/r02ah/pytest.log | cut -c1-200
for wg in 0 256; do
PT_DEBUG_CLOCK=1 python bench.py --config 5 --steps 256 --workgroup $wg --no-cpu-baseline 2> gpurun_out/r02ah/err_$wg.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('wg', $wg, round(json.loads(l)['value']))
"
grep -m1 "ptamd" gpurun_out/r02ah/err_$wg.txt | cut -c1-250
done
PT_DEBUG_CLOCK=1 python bench.py --config 5 --steps 256 --geom-path 8 --no-cpu-baseline 2> gpurun_out/r02ah/err_g8.txt | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('path8', round(json.loads(l)['value']))
"
grep -m1 "ptamd" gpurun_out/r02ah/err_g8.txt | cut -c1-250
