# the driver's 20-step call as 2, 3 or 4 batches (--batch 10 / 7 / 5): does an extra batch hide the accumulates that a 10 + 10 call leaves exposed at its end?
O=gpurun_out/$1; mkdir -p $O
for i in 1 2 3; do for b in 10 7 5; do
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 --batch $b > $O/drv_b${b}_$i.json 2>>$O/err.txt
done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/drv_b*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append((round(j['value']), j['config']['timed_batches']))
for k,v in sorted(r.items()): print(k, v)
PY
