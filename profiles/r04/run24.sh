#!/bin/bash
# round 4: resident-path instances with direct lighting and scattering: suite, A/B; then the N = 2 rehearsal of bench.py (two ranks on one GPU over gloo)
set -o pipefail
OUT=gpurun_out/r04x; mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -k "not multi_device_gather" > $OUT/pytest.log 2>&1 || { tail -40 $OUT/pytest.log; exit 1; }
tail -1 $OUT/pytest.log
b() { local name=$1; shift
  timeout -k 10 300 env "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "FAILED $name"; tail -5 $OUT/$name.err; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['value']:9.0f} Mrb/s  shadow {d['shadow_rays']:12d} frac {d['roofline']['frac']:.3f} resident {d['config']['resident_paths']}")
PY
}
for rep in 1 2; do
b dl_off_$rep PT_X=0 python bench.py --no-cpu-baseline --direct-light --steps 128 --resident -1
b dl_res_$rep PT_X=0 python bench.py --no-cpu-baseline --direct-light --steps 128 --resident 1
b dl5_off_$rep PT_X=0 python bench.py --no-cpu-baseline --direct-light --config 5 --steps 128 --resident -1
b dl5_res_$rep PT_X=0 python bench.py --no-cpu-baseline --direct-light --config 5 --steps 128 --resident 1
b dl3_off_$rep PT_X=0 python bench.py --no-cpu-baseline --direct-light --config 3 --steps 128 --resident -1
b dl3_res_$rep PT_X=0 python bench.py --no-cpu-baseline --direct-light --config 3 --steps 128 --resident 1
done
echo "== rehearsal N=2 (gloo, one GPU)"
PT_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/rehearsal_weak_n2.json 2> $OUT/rehearsal_n2.err || { tail -20 $OUT/rehearsal_n2.err; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04x/rehearsal_weak_n2.json") if l.startswith("{")][-1])
print("rehearsal", round(d["value"]), d["n_gpus"], d["config"]["gather_check"], d["config"]["parallelism"], d.get("value_pipelined_gather") and round(d["value_pipelined_gather"]))
PY
PT_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 5 --config 4 --scaling strong --bands > $OUT/rehearsal_strong_c4_n2.json 2>> $OUT/rehearsal_n2.err || { tail -20 $OUT/rehearsal_n2.err; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04x/rehearsal_strong_c4_n2.json") if l.startswith("{")][-1])
print("rehearsal strong bands", round(d["value"]), d["n_gpus"], d["config"]["gather_check"], d["config"]["parallelism"])
PY
