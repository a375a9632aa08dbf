O=gpurun_out/${1:-r03y}
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "sequences" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
for i in 1 2; do
for s in 1 2 3; do python bench.py --no-cpu-baseline --sequences $s > $O/c2_s${s}_$i.json 2>>$O/err.txt; done
for b in 4 8; do python bench.py --no-cpu-baseline --batch $b > $O/c2_b${b}_$i.json 2>>$O/err.txt; done
for w in 5 7; do PT_MAX_WG_PER_CU=$w python bench.py --no-cpu-baseline > $O/c2_w${w}_$i.json 2>>$O/err.txt; done
for s in 2 3; do python bench.py --no-cpu-baseline --config 5 --steps 128 --sequences $s > $O/c5_s${s}_$i.json 2>>$O/err.txt; done
done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/c*_[12].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in sorted(r.items()): print(f"{k:12s} {v}")
PY
