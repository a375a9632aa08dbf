# 832-thread workgroups of the batched walk (13 waves, 72 KB of LDS, kernels capped at 72 VGPRs): two per CU = 6.5 waves per SIMD; with one and two sequences
O=gpurun_out/$1; mkdir -p $O; V=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_wg896/libptamd.so
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --config 5 --steps 128 --sequences 1 > $O/c5_wg512_s1_$i.json 2>>$O/err.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=2 PT_DEBUG_CLOCK=1 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 832 --sequences 1 > $O/c5_wg832_s1_$i.json 2>$O/err_832_$i.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=1 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 832 --sequences 1 > $O/c5_wg832_1percu_s1_$i.json 2>>$O/err.txt
  python bench.py --no-cpu-baseline --config 5 --steps 128 > $O/c5_wg512_s2_$i.json 2>>$O/err.txt
  PT_LIBPTAMD=$V PT_MAX_WG_PER_CU=2 python bench.py --no-cpu-baseline --config 5 --steps 128 --workgroup 832 > $O/c5_wg832_s2_$i.json 2>>$O/err.txt
done
grep -h "launch:" $O/err_832_1.txt | sort | uniq
python - <<PY
import json,glob,collections
r=collections.defaultdict(list)
for f in sorted(glob.glob("$O/c5*.json")):
    j=json.loads(open(f).read().strip().splitlines()[-1]); r[f.split('/')[-1][:-7]].append(round(j['value']))
for k,v in sorted(r.items()): print(f"{k:24s} {v}")
PY
