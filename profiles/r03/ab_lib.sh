# A/B of two library builds on one box: $1 = output tag, $2 = baseline lib dir (under project3-pathtracer_amd/), $3.. = bench args
O=gpurun_out/$1; B=$GRAFT_REPO_ROOT/project3-pathtracer_amd/$2/libptamd.so; shift; shift
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for i in 1 2 3; do
  PT_LIBPTAMD=$B python bench.py --no-cpu-baseline "$@" > $O/base_$i.json 2>>$O/err.txt
  python bench.py --no-cpu-baseline "$@" > $O/new_$i.json 2>>$O/err.txt
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*_[123].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']))
PY
