set -e
O=gpurun_out/r03a
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "sequences or batching or config1" > $O/pytest.log 2>&1
echo "pytest ok"
for i in 1 2 3; do for s in 1 2; do
  PT_SEQUENCES=$s python bench.py --no-cpu-baseline > $O/bench_c2_s${s}_$i.json 2>$O/err_c2_s${s}_$i.txt
done; done
for s in 1 2; do
  PT_SEQUENCES=$s python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_c2_drv_s${s}.json 2>>$O/err.txt
  PT_SEQUENCES=$s python bench.py --no-cpu-baseline --config 3 --steps 256 > $O/bench_c3_s${s}.json 2>>$O/err.txt
  PT_SEQUENCES=$s python bench.py --no-cpu-baseline --config 5 --steps 128 > $O/bench_c5_s${s}.json 2>>$O/err.txt
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03a/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']), round(j['roofline']['frac'],4))
    except Exception as e: print(f, 'ERR', e)
PY
