O=gpurun_out/${1:-r03rep}
mkdir -p $O
for i in 1 2 3 4; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/drv_$i.json 2>>$O/err.txt; done
python bench.py --no-cpu-baseline > $O/def_1.json 2>>$O/err.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv_cpu.json 2>>$O/err.txt
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']), j['value_cold'] and round(j['value_cold']), round(j['roofline']['frac'],3), j['warmup_settle']['extra_untimed_iterations'], round(j['total_ms'],2))
PY
