#!/bin/bash
# round 4: the driver-shaped run right behind a run that ends with ~20 s of CPU-only work (the oracle leg): does the settle phase reach the clock?
OUT=gpurun_out/r04w; mkdir -p $OUT
make -C project3-pathtracer_amd/csrc OUT=../lib_dbg EXTRA_HIPFLAGS=-DPT_DEBUG_BOUNDS=1 -j16 > $OUT/build_dbg.log 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/drv_first.json 2> $OUT/err.txt
python bench.py --steps 20 --warmup 5 > $OUT/drv_second.json 2>> $OUT/err.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/drv_third.json 2>> $OUT/err.txt
python bench.py > $OUT/c2.json 2>> $OUT/err.txt
python bench.py --steps 20 --warmup 5 > $OUT/drv_fourth.json 2>> $OUT/err.txt
python - <<'PY'
import json,glob
for f in ["drv_first","drv_second","drv_third","c2","drv_fourth"]:
    d=json.loads(open(f"gpurun_out/r04w/{f}.json").read().strip().splitlines()[-1]); print(f, round(d['value']), 'cold', round(d['value_cold']), 'frac', round(d['roofline']['frac'],3), d['warmup_settle'])
PY
