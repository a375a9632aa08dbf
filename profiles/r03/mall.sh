O=gpurun_out/${1:-r03z}
mkdir -p $O
for i in 1 2; do for l in lib; do for cfg in "1 4" "2 4" "2 2" "4 2" "4 4" "16 2"; do set -- $cfg
  PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/$l/libptamd.so python bench.py --no-cpu-baseline --batch $1 --sequences $2 > $O/${l}_b$1_s$2_$i.json 2>>$O/err.txt
done; done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/lib*_[12].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in sorted(r.items()): print(f"{k:22s} {v}")
PY
