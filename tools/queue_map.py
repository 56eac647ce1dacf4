#!/usr/bin/env python3
"""stream -> hardware queue map of a rocprofv3 --kernel-trace CSV (Queue_Id / Stream_Id columns) with the kernels that identify each stream"""
import csv, sys
from collections import defaultdict, Counter
d = defaultdict(Counter)
for r in csv.DictReader(open(sys.argv[1])):
    d[(r["Stream_Id"], r["Queue_Id"])][r["Kernel_Name"].split("(")[0].replace("void ", "")[-36:]] += 1
for (s, q), v in sorted(d.items(), key=lambda kv: int(kv[0][0])):
    print(f"stream {s:>3} -> queue {q}: {sum(v.values()):5d} kernels  {', '.join(k for k, _ in v.most_common(3))}")
