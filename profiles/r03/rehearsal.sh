# bench.py's N > 1 path rehearsed with 2 and 4 ranks sharing ONE GPU over gloo (code path only: the values say nothing about speed)
O=gpurun_out/${1:-r03j}
mkdir -p $O
export PT_BENCH_REHEARSAL=1
for n in 2 4; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 20 --warmup 5 > $O/rehearsal_weak_n$n.json 2> $O/err_weak_n$n.txt; echo "weak n=$n rc=$?"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --config 4 --scaling strong --steps 16 --warmup 4 > $O/rehearsal_strong_c4_n2.json 2> $O/err_strong.txt; echo "strong rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --bands --steps 8 --warmup 2 > $O/rehearsal_bands_n2.json 2> $O/err_bands.txt; echo "bands rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/rehearsal_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith('{')][-1]); print(f.split('/')[-1], round(j['value']), round(j['value_pipelined_gather']), j['config']['gather_check'], j['config']['workload'][:60])
    except Exception as e: print(f, 'ERR', e)
PY
tail -5 $O/err_weak_n2.txt
