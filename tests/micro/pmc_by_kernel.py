"""Per-kernel summary of a rocprofv3 --pmc run: python tests/micro/pmc_by_kernel.py <dir-with-*_counter_collection.csv> [substr ...]
FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); counters are in KiB."""
import csv, glob, sys, collections
d = sys.argv[1]
subs = sys.argv[2:]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
agg = collections.OrderedDict()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if subs and not any(s in k for s in subs):
            continue
        key = (k[:60], r["Grid_Size"], r["Counter_Name"])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for (k, g, c), (n, v) in agg.items():
    mul = 2.0 if c == "FETCH_SIZE" else 1.0
    print(f"{c:12s} {v * mul * 1024 / n / 1e9:8.3f} GB/launch  x{n:<3d} grid {g:>8s}  {k}")
