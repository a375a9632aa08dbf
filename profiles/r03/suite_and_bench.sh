# full GPU suite + the driver's bench command + default bench, one box
O=gpurun_out/${1:-r03c}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd_20_5.json 2> $O/bench.err && python bench.py --no-cpu-baseline > $O/bench_config2.json 2>> $O/bench.err
tail -c 600 $O/bench.err
python - <<PY
import json
for n in ("bench_driver_cmd_20_5","bench_config2"):
    j=json.loads(open("$O/"+n+".json").read().strip().splitlines()[-1])
    print(n, round(j["value"]), "cold", j["value_cold"] and round(j["value_cold"]), "frac", round(j["roofline"]["frac"],4), "alone", round(j["roofline"]["kernel_alone"]["frac"],4), "traffic/bytes", j["roofline"]["traffic"]/j["roofline"]["bytes_per_launch"], j["valu_roofline"]["executed"], j.get("vs_cpu_baseline"))
PY
