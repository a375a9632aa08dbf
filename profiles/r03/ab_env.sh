# A/B of one library under two environments on one box: $1 = tag, $2 = "VAR=value" for the B side (A = unset), $3.. = bench args
O=gpurun_out/$1; ENVB=$2; shift; shift
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline "$@" > $O/a_$i.json 2>>$O/err.txt
  env $ENVB python bench.py --no-cpu-baseline "$@" > $O/b_$i.json 2>>$O/err.txt
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/[ab]_[123].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']), round(j['roofline']['frac'],3))
PY
