#!/bin/bash
# round 4: the checks the driver runs at round end -- GPU suite, smoke, bench.py with the driver's flags
set -o pipefail
OUT=gpurun_out/r04y; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04y/bench_driver.json").read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['frac'],3), d['cpu_baseline']['value'], d['config']['resident_paths'])
PY
