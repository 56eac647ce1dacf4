#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 --kernel-trace CSV (two-stream run): consecutive segments of the step with
the kernels that cover them, classified as BIG (a kernel of >= 100 us is running) or SMALL (only short kernels: latency-bound
chains that leave most of the chip idle).
  python tools/step_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sn_wt_u_kernel" in r["Kernel_Name"]]
lo, hi = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in step)
iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]) for r in step]
ev = sorted(set([a for a, _, _ in iv] + [b for _, b, _ in iv]))
big_t = small_t = idle_t = 0
segs = []
for a, b in zip(ev, ev[1:]):
    act = [(s, e, n) for s, e, n in iv if s <= a and e >= b]
    if not act:
        idle_t += b - a; kind = "IDLE"
    elif any(e - s >= 100000 for s, e, _ in act):
        big_t += b - a; kind = "BIG"
    else:
        small_t += b - a; kind = "SMALL"
    if segs and segs[-1][0] == kind:
        segs[-1][2] = b
    else:
        segs.append([kind, a, b])
print(f"step wall {(t1 - t0) / 1e6:.3f} ms: BIG {big_t / 1e6:.3f}  SMALL-only {small_t / 1e6:.3f}  idle {idle_t / 1e6:.3f}; kernels {len(step)}")
for kind, a, b in segs:
    if (b - a) >= 60000:
        names = sorted({n for s, e, n in iv if s < b and e > a})
        print(f"  {kind:5s} {(a - t0) / 1e6:7.3f} .. {(b - t0) / 1e6:7.3f} ms ({(b - a) / 1e3:7.1f} us)  {', '.join(names)[:150]}")
