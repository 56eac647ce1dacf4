#!/usr/bin/env python3
"""Per-stream listing of one step of a rocprofv3 --kernel-trace CSV: start (ms from the step's first kernel), duration, stream, kernel.
   python tools/step_streams.py <kernel_trace.csv> [t0_ms t1_ms]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sn_wt_u_kernel" in r["Kernel_Name"]]
lo, hi = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
t0 = int(rows[lo]["Start_Timestamp"])
a = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
b = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
streams = {}
for r in rows[lo:hi]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6
    sid = streams.setdefault(r["Stream_Id"], len(streams))
    if e >= a and s <= b:
        print(f"{s:8.3f} {1e3 * (e - s):7.1f}us  s{sid} {'    ' * sid}{r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]}")
