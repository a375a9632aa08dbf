# instruction counts of the later-bounce pair kernel with and without two chunks in flight (PMC pass sq1 on both libraries, config 2)
O=gpurun_out/$1; mkdir -p $O
python3 profiles/collect_pmc.py --tag ${1}_pipe --passes sq1 > $O/pmc_pipe.log 2>&1
PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_nopipe/libptamd.so python3 profiles/collect_pmc.py --tag ${1}_nopipe --passes sq1 > $O/pmc_nopipe.log 2>&1
python3 - <<PY
import json
for t in ("pipe","nopipe"):
    s=json.load(open("gpurun_out/pmc_${1}_%s/summary.json"%t))
    for kn,k in s["kernels"].items():
        if "k_bounce" in kn:
            pm=k.get("pmc",{})
            print(t, kn, {n: round(v["sum"]/1e6,1) for n,v in pm.items()}, "dispatches", pm.get("SQ_WAVES",{}).get("dispatches"))
PY
