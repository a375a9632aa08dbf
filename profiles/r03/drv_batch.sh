O=gpurun_out/${1:-r03drv}
mkdir -p $O
for i in 1 2 3; do for b in 0 10 7 5 4; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --batch $b > $O/b${b}_$i.json 2>>$O/err.txt
done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/b*_[123].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in sorted(r.items()): print(f"{k:8s} {v}")
PY
