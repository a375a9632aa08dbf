#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace CSV into the figure bench.py's roofline block uses.

    python3 profiles/trace_union.py <..._kernel_trace.csv> [substring of the kernel name, default k_bounce]

With two launch sequences in flight (pt_options.sequences) a bounce launch of one sequence runs beside a launch of the
other: the durations a kernel trace lists overlap, their sum exceeds the time the GPU spent on them, and
`TotalDurationNs / Calls` of the --stats table is a launch's own lifetime, not the rate launches complete at.  This script
merges the [start, end) intervals of the matching dispatches:
    sum_ms     sum of the dispatches' own durations        (what --stats reports)
    union_ms   time during which at least one of them ran
    eff_ms     union_ms / dispatches = avg_launch_ms of bench.py's roofline block (bytes_per_launch / eff_ms = achieved)
    overlap    sum_ms / union_ms (1.0 = never two at once)
"""
import csv
import json
import sys


def main():
    path = sys.argv[1]
    key = sys.argv[2] if len(sys.argv) > 2 else "k_bounce"
    iv = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if key in row["Kernel_Name"]:
                iv.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
    iv.sort()
    if not iv:
        print(json.dumps({"kernel": key, "dispatches": 0}))
        return
    total = sum(e - s for s, e in iv)
    union, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            union += ce - cs
            cs, ce = s, e
        elif e > ce:
            ce = e
    union += ce - cs
    n = len(iv)
    print(json.dumps({"kernel": key, "dispatches": n, "sum_ms": total / 1e6, "union_ms": union / 1e6,
                      "own_avg_ms": total / n / 1e6, "eff_ms": union / n / 1e6, "overlap": total / union}))


if __name__ == "__main__":
    main()
