#!/bin/bash
# round 4, closing: the N = 2 rehearsal of bench.py on the final build (two ranks on one GPU over gloo: code path only), then more fuzz on the final kernels
OUT=gpurun_out/r04zp; mkdir -p $OUT
echo "== rehearsal N=2 (gloo, one GPU)"
PT_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/rehearsal_weak_n2.json 2> $OUT/rehearsal_n2.err || { tail -20 $OUT/rehearsal_n2.err; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04zp/rehearsal_weak_n2.json") if l.startswith("{")][-1])
print("rehearsal", round(d["value"]), d["n_gpus"], d["config"]["gather_check"], d["config"]["parallelism"], d.get("value_pipelined_gather") and round(d["value_pipelined_gather"]))
PY
PT_BENCH_REHEARSAL=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 5 --config 4 --scaling strong --bands > $OUT/rehearsal_strong_c4_n2.json 2>> $OUT/rehearsal_n2.err || { tail -20 $OUT/rehearsal_n2.err; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04zp/rehearsal_strong_c4_n2.json") if l.startswith("{")][-1])
print("rehearsal strong bands", round(d["value"]), d["n_gpus"], d["config"]["gather_check"], d["config"]["parallelism"])
PY
DBG=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_dbg/libptamd.so
timeout -k 10 560 python tests/fuzz_gpu.py 60000 4000000 > $OUT/fuzz.log 2>&1; tail -1 $OUT/fuzz.log; grep -m5 MISMATCH $OUT/fuzz.log
PT_LIBPTAMD=$DBG PT_DEBUG_BOUNDS=1 timeout -k 10 400 python tests/fuzz_gpu.py 25000 4100000 > $OUT/fuzz_bounds.log 2>&1; tail -1 $OUT/fuzz_bounds.log; echo "violations: $(grep -c 'BOUNDS violation' $OUT/fuzz_bounds.log)"; grep -m5 "MISMATCH\|BOUNDS" $OUT/fuzz_bounds.log
PT_FUZZ_SCENES=extreme timeout -k 10 300 python tests/fuzz_gpu.py 15000 4200000 > $OUT/fuzz_extreme.log 2>&1; tail -1 $OUT/fuzz_extreme.log; grep -m5 MISMATCH $OUT/fuzz_extreme.log
exit 0
