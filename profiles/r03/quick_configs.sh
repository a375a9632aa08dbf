O=gpurun_out/${1:-r03s}
mkdir -p $O
python bench.py --no-cpu-baseline > $O/c2.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/c2_drv.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --config 3 --steps 256 > $O/c3.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --config 4 --steps 64 > $O/c4.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --config 5 --steps 256 > $O/c5.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --config 1 --steps 256 > $O/c1.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --direct-light > $O/c2_nee.json 2>>$O/err.txt
python bench.py --no-cpu-baseline --rotat degrees > $O/c2_deg.json 2>>$O/err.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']), j['value_cold'] and round(j['value_cold']), round(j['roofline']['frac'],3), round(j['roofline']['kernel_alone']['frac'],3))
PY
