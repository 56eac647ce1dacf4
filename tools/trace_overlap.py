#!/usr/bin/env python3
"""Timeline view of one training step from a rocprofv3 --kernel-trace CSV: wall span of the step, summed kernel time,
and how much of every kernel class ran while another kernel was also running (stream-level overlap).
  python tools/trace_overlap.py <kernel_trace.csv>"""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sn_wt_u_kernel" in r["Kernel_Name"]]
lo, hi = marks[len(marks) // 2], marks[len(marks) // 2 + 1]     # a step from the middle of the timed region
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, kernels {len(step)}, summed kernel time {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e6:.3f} ms")
ev = []
for i, r in enumerate(step):
    ev.append((int(r["Start_Timestamp"]), 1, i)); ev.append((int(r["End_Timestamp"]), -1, i))
ev.sort()
active = set(); last = None; overl = collections.Counter(); tot = collections.Counter()
for t, d, i in ev:
    if last is not None and active:
        dt = t - last
        for j in active:
            k = step[j]["Kernel_Name"].split("(")[0][:34]
            tot[k] += dt
            if len(active) > 1:
                overl[k] += dt
    if d > 0: active.add(i)
    else: active.discard(i)
    last = t
for k, v in tot.most_common(14):
    print(f"  {k:36s} {v / 1e6:7.3f} ms, of which overlapped with another kernel {overl[k] / 1e6:7.3f} ms")
