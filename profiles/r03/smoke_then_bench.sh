O=gpurun_out/${1:-r03sb}
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/a_alone.json 2>>$O/err.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/b_after_smoke.json 2>>$O/err.txt
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/c_after_suite.json 2>>$O/err.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/d_after_smoke_cpu.json 2>>$O/err.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']), j['value_cold'] and round(j['value_cold']), round(j['roofline']['frac'],3))
PY
