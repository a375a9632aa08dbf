O=gpurun_out/${1:-r03t1}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py --no-cpu-baseline > $O/c2.json 2>$O/err.txt; python -c "
import json; j=json.loads(open('$O/c2.json').read().strip().splitlines()[-1]); print(round(j['value']))"
