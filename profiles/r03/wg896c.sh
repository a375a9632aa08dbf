# were two 896-thread workgroups (78 KB of LDS each) ever resident together?  the same run with ONE workgroup per CU, and 832-thread ones (72 KB each) for comparison
O=gpurun_out/$1; mkdir -p $O; V=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_wg896/libptamd.so
for i in 1 2; do
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=1 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 896 --sequences 1 > $O/c5_wg896_1percu_s1_$i.json 2>>$O/err.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=2 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 896 --sequences 1 > $O/c5_wg896_2percu_s1_$i.json 2>>$O/err.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=1 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 512 --sequences 1 > $O/c5_wg512_1percu_s1_$i.json 2>>$O/err.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=2 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 512 --sequences 1 > $O/c5_wg512_2percu_s1_$i.json 2>>$O/err.txt
  PT_LIBPTAMD=$V python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 512 --sequences 1 > $O/c5_wg512_3percu_s1_$i.json 2>>$O/err.txt
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/c5*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(j['value']))
PY
