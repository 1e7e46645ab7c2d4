#!/usr/bin/env python3
"""Per-kernel durations from a rocprofv3 --kernel-trace output directory (csv or rocpd database): tools/kernel_times.py <dir> [name filter]"""
import csv, glob, sqlite3, sys, collections
d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "ptx")
rows = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
for f in glob.glob(d + "/**/*.db", recursive=True):
    for n, s, e in sqlite3.connect(f).execute("select name, start, end from kernels"):
        rows.append((s, n, (e - s) / 1e3))
rows.sort()
agg = collections.OrderedDict()
for _, n, us in rows:
    if flt in n:
        k = n.split("(")[0][-48:]
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += us
        if "-v" in sys.argv: print(f"{k:50s} {us:10.1f} us")
for k, (c, us) in agg.items(): print(f"{k:50s} calls {c:4d}  total {us/1e3:9.3f} ms  avg {us/c:9.1f} us")
