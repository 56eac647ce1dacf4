#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel stats / kernel trace / PMC counter collection) of a bench.py run into the
small per-step summaries committed under profiles/.
  python tools/summarize_profile.py stats  <kernel_stats.csv> <steps_in_run | 0 = count sn_wt_u_kernel launches>
  python tools/summarize_profile.py pmc    <counter_collection.csv>      # FETCH_SIZE or WRITE_SIZE pass
  python tools/summarize_profile.py traffic <fetch.csv> <write.csv>      # JSON read by bench.py (roofline.traffic)
Counter units: rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts wide coalesced
reads at half their bytes (MI355X_MICROARCH.md, HBM section) -> reads are multiplied by 2."""
import collections
import csv
import sys


SETUP = ("transpose_kernelIf", "at::native", "make_copies_kernel")     # dataset generation / conversion, first weight copies


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    if steps <= 0:   # every training forward starts its power iteration with exactly one sn_wt_u_kernel launch (the prefetched augmentation runs in several launches in the middle of a step)
        steps = sum(int(r["Calls"]) for r in rows if "sn_wt_u_kernel" in r["Name"])
    setup = [r for r in rows if any(k in r["Name"] for k in SETUP)]
    rows = [r for r in rows if r not in setup]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"| kernel | launches/step | ms/step | avg us | % of step kernels |\n|---|---|---|---|---|")
    for r in rows[:28]:
        print(f"| `{r['Name'][:80]}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / tot * 100:.1f} |")
    print(f"\ntotal kernel time {tot / 1e6 / steps:.2f} ms/step over {sum(int(r['Calls']) for r in rows) / steps:.0f} launches/step "
          f"(run of {steps} steps incl. warm-up and the kernel-timing pass); one-off set-up kernels (synthetic dataset generation and "
          f"layout conversion, first weight copies) excluded: {sum(float(r['TotalDurationNs']) for r in setup) / 1e6:.1f} ms in total")


def pmc(path):
    rows = list(csv.DictReader(open(path)))
    name = rows[0]["Counter_Name"]
    corr = 2.0 if name == "FETCH_SIZE" else 1.0
    idx = [i for i, x in enumerate(rows) if "sn_wt_u_kernel" in x["Kernel_Name"]]     # one per training step, first kernel of its forward pass
    step = rows[idx[-2]: idx[-1]]
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


def step_rows(path):
    rows = list(csv.DictReader(open(path)))
    idx = [i for i, x in enumerate(rows) if "sn_wt_u_kernel" in x["Kernel_Name"]]
    return rows[0]["Counter_Name"], rows[idx[-2]: idx[-1]]


def traffic(fetch_csv, write_csv):
    """JSON for bench.py's roofline.traffic: HBM bytes (reads x2 corrected, writes) and launches of each GEMM kernel
    over one training step of the profiled run."""
    import json
    out = {}
    for path in (fetch_csv, write_csv):
        name, step = step_rows(path)
        corr = 2.0 if name == "FETCH_SIZE" else 1.0
        for x in step:
            k = x["Kernel_Name"]
            fam = next((n for n in ("gemm_nt_t256_kernel", "t256_reduce_kernel", "gemm_nt_wide64p_kernel", "gemm_nt_reduce", "gemm_nt_kernel",
                                    "gemm_tn_t256_kernel", "gemm_tn_w2_kernel", "gemm_tn_kernel", "sum_slabs_kernel", "adamw_sn_kernel") if n in k), None)
            if fam is None:
                continue
            aux = fam in ("t256_reduce_kernel",)    # the 256x256 kernel's split-K combine: bytes of its class, not a launch of it
            if fam in ("gemm_tn_kernel", "gemm_tn_w2_kernel"):
                fam = "gemm_tn_t256_kernel"    # one weight-gradient class (bench.py's roofline_gemm_tn covers the three kernels)
            if fam == "t256_reduce_kernel":
                fam = "gemm_nt_t256_kernel"
            d = out.setdefault(fam, {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0})
            d["fetch_bytes" if name == "FETCH_SIZE" else "write_bytes"] += float(x["Counter_Value"]) * 1024 * corr
            if name == "FETCH_SIZE" and not aux:
                d["launches"] += 1
    out["note"] = ("one training step of bench.py (batch 16, bf16) under rocprofv3 --pmc; FETCH_SIZE x2 (gfx950 correction, "
                   "MI355X_MICROARCH.md HBM section), WRITE_SIZE x1, counters in KiB")
    print(json.dumps(out, indent=1))


def traffic_lc(fetch_csv, write_csv):
    """Latent-conditioner bench (bench.py --workload lc): bytes of the two GEMM operator classes over the whole profiled run
    (every step is the same), keys lc_gemm_nt / lc_gemm_tn; a launch = one main kernel (its split-K combine adds bytes only)."""
    import json
    out = {}
    for path in (fetch_csv, write_csv):
        rows = list(csv.DictReader(open(path)))
        name = rows[0]["Counter_Name"]
        corr = 2.0 if name == "FETCH_SIZE" else 1.0
        for x in rows:
            k = x["Kernel_Name"]
            if "gemm_tn" in k or "sum_slabs" in k or "stem_conv_dw" in k or "stem_dw_finalize" in k:
                fam, main = "lc_gemm_tn", ("gemm_tn" in k or "stem_conv_dw" in k)
            elif "gemm_nt" in k or "t256_reduce" in k or "stem_conv_fwd" in k or "stem_stats_finalize" in k:
                fam, main = "lc_gemm_nt", ("reduce" not in k and "finalize" not in k)
            else:
                continue
            d = out.setdefault(fam, {"fetch_bytes": 0.0, "write_bytes": 0.0, "launches": 0})
            d["fetch_bytes" if name == "FETCH_SIZE" else "write_bytes"] += float(x["Counter_Value"]) * 1024 * corr
            if name == "FETCH_SIZE" and main:
                d["launches"] += 1
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]))
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic_lc":
        traffic_lc(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2])
