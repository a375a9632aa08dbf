#!/bin/bash
# round 4: kernel timeline of the default bench run (two sequences, resident paths)
OUT=gpurun_out/r04h; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 128 > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/bench.err
cd $GRAFT_REPO_ROOT
CSV=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
echo $CSV
python3 profiles/timeline.py $CSV > $OUT/timeline.txt
head -60 $OUT/timeline.txt
python3 profiles/trace_union.py $CSV k_bounce
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep the merge small: the raw trace is large
python3 - <<PY
import csv,sys
rows=list(csv.DictReader(open("$CSV")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)
with open("$OUT/trace_tail.csv","w") as f:
    w=csv.DictWriter(f,fieldnames=["Queue_Id","Start_Timestamp","End_Timestamp","Kernel_Name"]); w.writeheader()
    for r in rows[max(0,n-400):]: w.writerow({k:r[k] for k in w.fieldnames})
PY
rm -rf $OUT/trace
