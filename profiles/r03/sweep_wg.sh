O=gpurun_out/${1:-r03l}
mkdir -p $O
for i in 1 2; do for wg in 128 256 512; do
  python bench.py --no-cpu-baseline --workgroup $wg > $O/c2_wg${wg}_$i.json 2>>$O/err.txt
done; done
for w in 6 7 8; do PT_MAX_WG_PER_CU=$w python bench.py --no-cpu-baseline > $O/c2_w${w}_1.json 2>>$O/err.txt; done
for wg in 256 512; do for w in 2 3 4; do
  PT_MAX_WG_PER_CU=$w python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup $wg > $O/c5_wg${wg}_w${w}_1.json 2>>$O/err.txt
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']))
PY
