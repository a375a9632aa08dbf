O=gpurun_out/${1:-r03d}
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "sequences or batching or resume or leak or large_mesh" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
PT_DEBUG_CLOCK=1 python profiles/r03/cold_start.py 20 > $O/cold_plain.txt 2> $O/cold_plain_clk.txt
PT_PRETOUCH=1 PT_DEBUG_CLOCK=1 python profiles/r03/cold_start.py 20 > $O/cold_pretouch.txt 2> $O/cold_pretouch_clk.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_drv.json 2>$O/err.txt
cat $O/cold_plain.txt; grep MHz $O/cold_plain_clk.txt | head -40
echo ---- pretouch; head -12 $O/cold_pretouch.txt
python - <<PY
import json
j=json.loads(open("$O/bench_drv.json").read().strip().splitlines()[-1]); print(round(j["value"]), j["value_cold"])
PY
