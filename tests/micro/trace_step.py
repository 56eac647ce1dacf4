#!/usr/bin/env python3
"""One training step out of a rocprofv3 --kernel-trace CSV, kernel by kernel: start (us from the step's first kernel), duration, queue, name.
  python tests/micro/trace_step.py <kernel_trace.csv> [min_us] [step_index]
The step runs from one latent_fwd_kernel to the next (one per forward pass)."""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
marks = [i for i, r in enumerate(rows) if "latent_fwd" in r["Kernel_Name"]]
k = int(sys.argv[3]) if len(sys.argv) > 3 else len(marks) // 2
step = rows[marks[k]:marks[k + 1]]
t0 = int(step[0]["Start_Timestamp"])
print(f"step {k}: {(int(rows[marks[k + 1]]['Start_Timestamp']) - t0) / 1e3:.1f} us, {len(step)} kernels")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (e - s) / 1e3 >= min_us:
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:48]
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} q{r['Queue_Id']} {n}")
