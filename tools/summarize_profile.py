#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel stats / kernel trace / PMC counter collection) of a bench.py run into the
small per-step summaries committed under profiles/.
  python tools/summarize_profile.py stats  <kernel_stats.csv> <steps_in_run | 0 = count AdamW launches>
  python tools/summarize_profile.py pmc    <counter_collection.csv>      # FETCH_SIZE or WRITE_SIZE pass
Counter units: rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts wide coalesced
reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> reads are multiplied by 2."""
import collections
import csv
import sys


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    if steps <= 0:   # every training step launches the fused AdamW kernel exactly once
        steps = sum(int(r["Calls"]) for r in rows if "adamw_kernel<true" in r["Name"])
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"| kernel | launches/step | ms/step | avg us | % |\n|---|---|---|---|---|")
    for r in rows[:28]:
        print(f"| `{r['Name'][:80]}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    print(f"\ntotal kernel time {tot / 1e6 / steps:.2f} ms/step over {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step "
          f"(run of {steps} steps incl. warm-up and the kernel-timing pass)")


def pmc(path):
    rows = list(csv.DictReader(open(path)))
    name = rows[0]["Counter_Name"]
    corr = 2.0 if name == "FETCH_SIZE" else 1.0
    idx = [i for i, x in enumerate(rows) if x["Kernel_Name"].startswith("void adamw_kernel")]
    step = rows[idx[-2] + 1: idx[-1] + 1]
    agg, cnt, dur = collections.Counter(), collections.Counter(), collections.Counter()
    for x in step:
        k = x["Kernel_Name"].split("(")[0][:60]
        agg[k] += float(x["Counter_Value"]) * 1024 * corr
        cnt[k] += 1
        dur[k] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
    print(f"{name} (x{corr:g} gfx950 correction), one training step:\n\n| kernel | launches | GB/step | GB/launch | TB/s while running |\n|---|---|---|---|---|")
    for k, v in agg.most_common(12):
        print(f"| `{k}` | {cnt[k]} | {v / 1e9:.2f} | {v / 1e9 / cnt[k]:.3f} | {v / dur[k] / 1e3:.2f} |")
    print(f"\ntotal {sum(agg.values()) / 1e9:.1f} GB/step")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    else:
        pmc(sys.argv[2])
