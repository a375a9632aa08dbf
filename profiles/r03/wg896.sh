O=gpurun_out/${1:-r03x}
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "cloud or large or walk" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
for i in 1 2 3; do for wg in 512 896; do
  PT_DEBUG_CLOCK=1 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup $wg > $O/c5_wg${wg}_$i.json 2>$O/err_wg${wg}_$i.txt
done; done
grep -h "launch:" $O/err_wg896_1.txt $O/err_wg512_1.txt | sort | uniq
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/c5*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']))
PY
