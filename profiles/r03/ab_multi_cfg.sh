# like ab_multi.sh for several bench configurations: $1 = tag, $2 = "dir1 dir2", $3 = "cfgA|cfgB" bench argument sets separated by |
O=gpurun_out/$1; LIBS=$2; IFS='|' read -ra CFGS <<< "$3"
mkdir -p $O
k=0
for cfg in "${CFGS[@]}"; do k=$((k+1)); for i in 1 2 3; do for l in $LIBS; do
  PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/$l/libptamd.so python bench.py --no-cpu-baseline $cfg > $O/cfg${k}_${l}_$i.json 2>>$O/err.txt
done; done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/cfg*_[123].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in sorted(r.items()): print(f"{k:24s} {v}  mean {sum(v)/len(v):.0f}")
PY
