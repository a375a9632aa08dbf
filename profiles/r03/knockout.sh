O=gpurun_out/${1:-r03m}
LIBS="lib lib_ko1_200 lib_ko5_100 lib_ko2_50 lib_ko3_32 lib_ko4_200"
mkdir -p $O
for s in 2 1; do for i in 1 2; do for l in $LIBS; do
  PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/$l/libptamd.so python bench.py --no-cpu-baseline --sequences $s > $O/${l}_s${s}_$i.json 2>>$O/err.txt
done; done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/*_[12].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in sorted(r.items()): print(f"{k:18s} {v}  mean {sum(v)/len(v):.0f}")
PY
