#!/bin/bash
# round 4: full GPU suite after the multi-device gather rewrite, the stream guards, the allocation fallback and the serial-budget helper
set -o pipefail
OUT=gpurun_out/r04k; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -s -k "not fuzz" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
grep -E "strips=(0|8):" $OUT/pytest.log
for rep in 1 2; do python bench.py --no-cpu-baseline > $OUT/c2_$rep.json 2>$OUT/c2_$rep.err; python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/drv_$rep.json 2>$OUT/drv_$rep.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04k/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']), round(d['roofline']['frac'],3))
PY
