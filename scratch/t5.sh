for r in 1 2 3; do
for v in base new; do
if [ $v = base ]; then export PT_LIBPTAMD=$PWD/project3-pathtracer_amd/lib_base/libptamd.so; else unset PT_LIBPTAMD; fi
python bench.py --config 2 --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): print('$v', round(json.loads(l)['value']))
"
done
done
