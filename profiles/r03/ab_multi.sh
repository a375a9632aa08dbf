# A/B/C... of several library builds on one box, alternating: $1 = output tag, $2 = "dir1 dir2 ..." (under project3-pathtracer_amd/), $3.. = bench args
O=gpurun_out/$1; LIBS=$2; shift; shift
mkdir -p $O
for i in 1 2 3; do for l in $LIBS; do
  PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/$l/libptamd.so python bench.py --no-cpu-baseline "$@" > $O/${l}_$i.json 2>>$O/err.txt
done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/*_[123].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in r.items(): print(f"{k:12s} {v}  mean {sum(v)/len(v):.0f}")
PY
