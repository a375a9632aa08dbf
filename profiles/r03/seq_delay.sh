O=gpurun_out/${1:-r03sd}
mkdir -p $O
for i in 1 2 3; do for d in 0 300 700 1200; do
  PT_SEQ_DELAY_US=$d python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/drv_d${d}_$i.json 2>>$O/err.txt
  PT_SEQ_DELAY_US=$d python bench.py --no-cpu-baseline > $O/def_d${d}_$i.json 2>>$O/err.txt
done; done
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/*_[123].json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in sorted(r.items()): print(f"{k:14s} {v}")
PY
