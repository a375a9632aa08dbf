# option sweep used during round 1 (run on the GPU box): bash profiles/sweep_r01.sh
for args in "--geom-path 1" "--geom-path 2" "--geom-path 3" "--geom-path 3 --workgroup 128" "--geom-path 3 --workgroup 512" "--geom-path 3 --rotat degrees" "--geom-path 3 --scene cloud256.txt --depth 8 --rotat degrees" "--geom-path 1 --scene cloud256.txt --depth 8 --rotat degrees"; do
  echo "== $args"
  timeout -k 5 120 python bench.py --steps 128 --warmup 8 --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['value'],1),'Mrb/s', round(j['ms_per_step'],4),'ms/step', 'kernel-avg-us', round(j['roofline']['avg_launch_ms']*1e3,2), 'frac', round(j['roofline']['frac'],4))"
done
