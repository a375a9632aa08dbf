# 7 waves per SIMD for the batched walk, properly this time: 896-thread workgroups with the kernels capped at 72 VGPRs (launch bounds), two per CU.
# Variant build -DPT_WG896 (units 6, context).  $1 = tag
O=gpurun_out/$1; mkdir -p $O; V=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_wg896/libptamd.so
PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=2 timeout -k 10 300 python profiles/r03/wg896_check.py > $O/check.log 2>&1; echo "check rc=$?"; tail -3 $O/check.log
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --config 5 --steps 128 > $O/c5_wg512_$i.json 2>>$O/err.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=2 PT_DEBUG_CLOCK=1 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 896 > $O/c5_wg896_$i.json 2>$O/err_wg896_$i.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=2 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 896 --sequences 1 > $O/c5_wg896s1_$i.json 2>>$O/err.txt
done
grep -h "launch:" $O/err_wg896_1.txt | sort | uniq
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/c5*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']))
PY
