python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): p=json.loads(l); print('n1', round(p['value']))
"
PT_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/reh2.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): p=json.loads(l); print('rehearsal n2', round(p['value']), p['n_gpus'], p['config']['parallelism'], p['config']['gather_ms_standalone'])
"
tail -3 gpurun_out/reh2.err
PT_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 4 --config 4 --scaling strong --steps 16 --warmup 5 --no-cpu-baseline 2> gpurun_out/reh4.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): p=json.loads(l); print('rehearsal n4 strong', round(p['value']), p['n_gpus'], p['scaling'], p['config']['gather_ms_standalone'])
"
tail -3 gpurun_out/reh4.err
