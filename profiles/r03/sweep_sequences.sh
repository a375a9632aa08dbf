# sweep: launch sequences in flight x workgroups per CU x batch, config 2 (and 5), one box
set -e
O=gpurun_out/r03b
mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --no-cpu-baseline $EXTRA > $O/$n.json 2>>$O/err.txt
}
for s in 1 2 3 4; do for w in 4 5 6; do
  EXTRA="" run c2_s${s}_w${w} PT_SEQUENCES=$s PT_MAX_WG_PER_CU=$w
done; done
for s in 2 3 4; do for b in 4 8; do
  EXTRA="--batch $b" run c2_s${s}_b${b} PT_SEQUENCES=$s
done; done
for s in 1 2 3 4; do
  EXTRA="--config 5 --steps 128" run c5_s${s} PT_SEQUENCES=$s
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03b/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']))
    except Exception as e: print(f, 'ERR', e)
PY
