O=gpurun_out/${1:-r03bvh}
mkdir -p $O
for c in 0 1 2; do
  PT_W4_COLLAPSE=$c PT_DEBUG_W4=1 PT_LIBPTAMD=$GRAFT_REPO_ROOT/project3-pathtracer_amd/lib_bvhdbg/libptamd.so python bench.py --no-cpu-baseline --config 5 --steps 32 --warmup 16 --settle-ms 0 > $O/dbg_$c.json 2> $O/dbg_$c.txt
  echo "collapse $c: $(grep 'batched walk' $O/dbg_$c.txt | tail -1)"
  for i in 1 2; do PT_W4_COLLAPSE=$c python bench.py --no-cpu-baseline --config 5 --steps 128 > $O/c5_$c_$i.json 2>>$O/err.txt; python -c "
import json; j=json.loads(open('$O/c5_$c_$i.json').read().strip().splitlines()[-1]); print('  ', round(j['value']))"; done
done
