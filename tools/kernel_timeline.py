#!/usr/bin/env python3
"""Timeline (start, duration, stream/queue) of the kernels in a rocprofv3 --kernel-trace csv directory: tools/kernel_timeline.py <dir> [first] [count]"""
import csv, glob, sys
d = sys.argv[1]; first = int(sys.argv[2]) if len(sys.argv) > 2 else 0; count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-30:], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
rows = [r for r in rows if "ptx" in r[2]]
t0 = rows[first][0]
for s, e, n, q, st in rows[first:first + count]:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q{q} s{st}  {n}")
