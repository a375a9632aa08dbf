#!/usr/bin/env python3
"""Print a stretch of a rocprofv3 --kernel-trace CSV as a timeline: per dispatch its queue, start and end (us, relative) and
short kernel name -- to see how the launch sequences' kernels overlap.   python3 profiles/timeline.py <csv> [first] [count]"""
import csv
import re
import sys


def short(name):
    m = re.search(r"k_bounce<(\d+), (true|false), (\d+), (\d+), (\d+)>", name)
    if m:
        wg, first, geom, comp, feat = m.groups()
        return "camera" if first == "true" else ("resident" if int(feat) & 8 else "bounce")
    for k in ("k_accumulate", "k_iter_begin", "k_iter_set", "k_iter_fold"):
        if k in name:
            return k[2:]
    return name[:24]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    t0 = int(rows[first]["Start_Timestamp"])
    by = {}
    for r in rows:
        d = by.setdefault(short(r["Kernel_Name"]), [0, 0])
        d[0] += 1
        d[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"# {k:12s} {n:6d} dispatches, own duration {t / n / 1e3:9.1f} us on average, {t / 1e6:9.2f} ms in all")
    for r in rows[first:first + count]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"q{r.get('Queue_Id', '?'):>3s} {s / 1e3:9.1f} .. {e / 1e3:9.1f} us  ({(e - s) / 1e3:8.1f})  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
