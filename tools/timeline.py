#!/usr/bin/env python3
"""Dispatch timeline of a few bench steps from a rocprofv3 --kernel-trace CSV: which kernels run side by side.
usage: tools/timeline.py <dir with *_kernel_trace.csv> [first dispatch index] [count]"""
import csv
import glob
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else -60
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = rows[first:][:count]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:24]
    bar = " " * int(s / 20e3) + "#" * max(1, int((e - s) / 20e3))
    print("%-24s q%-3s %8.1f -> %8.1f us (%6.1f)  |%s" % (name, r.get("Queue_Id", "?")[-3:], s / 1e3, e / 1e3, (e - s) / 1e3, bar))
