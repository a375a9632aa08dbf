# option sweep used during round 1 (run on the GPU box): bash profiles/sweep_r01.sh
for args in "--batch 1" "--batch 2" "--batch 4" "--batch 4 --workgroup 128" "--batch 4 --workgroup 512" "--batch 4 --geom-path 1" "--batch 4 --rotat degrees" "--batch 4 --compaction 2"; do
  echo "== $args"
  timeout -k 5 120 python bench.py --steps 128 --warmup 8 --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'Mrb/s', round(j['ms_per_step'],4),'ms/step', 'kernel-avg-us', round(j['roofline']['avg_launch_ms']*1e3,2), 'frac', round(j['roofline']['frac'],4))"
done
