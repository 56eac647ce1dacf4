#!/usr/bin/env python3
"""Assemble profiles/<tag>_summary.md (+ the raw files it cites) from one tools/profile_round2.sh output directory:
  python tools/write_round2_summary.py gpurun_out/r02a r02 ["free text appended under 'Notes'"]"""
import collections, csv, glob, json, os, shutil, subprocess, sys

O, TAG = sys.argv[1], sys.argv[2]
notes = sys.argv[3] if len(sys.argv) > 3 else ""
here = os.path.dirname(os.path.abspath(__file__))
P = os.path.join(os.path.dirname(here), "profiles")
run = lambda tool, *a: subprocess.run([sys.executable, os.path.join(here, tool), *a], capture_output=True, text=True, check=True).stdout
f = lambda rel: os.path.join(O, rel)
serial = run("summarize_profile.py", "stats", f("stats_serial/run_kernel_stats.csv"), "0")
overl = run("summarize_profile.py", "stats", f("stats/run_kernel_stats.csv"), "0")
fetch = run("summarize_profile.py", "pmc", f("fetch/run_counter_collection.csv"))
write = run("summarize_profile.py", "pmc", f("write/run_counter_collection.csv"))
traffic = json.loads(run("summarize_profile.py", "traffic", f("fetch/run_counter_collection.csv"), f("write/run_counter_collection.csv")))
traffic.update(json.loads(run("summarize_profile.py", "traffic_lc", f("lc_fetch/run_counter_collection.csv"), f("lc_write/run_counter_collection.csv"))))
timeline = run("step_timeline.py", f("stats/run_kernel_trace.csv")).strip()
bench = open(f("bench_line.json")).read().strip().splitlines()[-1]
lc_bench = open(f("lc_bench_line.json")).read().strip().splitlines()[-1]
b, lb = json.loads(bench), json.loads(lc_bench)
layers = [l[:170] for l in open(f("layers.log")).read().splitlines() if " us " in l and "TF/s" in l][:40]
# latent-conditioner kernel table (whole run / steps)
lc_rows = list(csv.DictReader(open(f("lc_stats/run_kernel_stats.csv"))))
lc_tot = sum(float(r["TotalDurationNs"]) for r in lc_rows)
lc_tab = "| kernel | launches | total ms | avg us | % |\n|---|---|---|---|---|\n" + "\n".join(
    f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / lc_tot * 100:.1f} |"
    for r in lc_rows[:22])
agg = collections.defaultdict(list)
for p in glob.glob(f("sq_*/run_counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        if "gemm_nt_t256_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
sq = "\n".join(f"{k:28s} {sorted(v)[len(v) // 2]:.3e}" for k, v in sorted(agg.items()))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f("sq_SQ_WAVE_CYCLES/run_kernel_trace.csv")))
       if "gemm_nt_t256_kernel" in r["Kernel_Name"]]
dur_us = sorted(dur)[len(dur) // 2] / 1e3 if dur else float("nan")
mf = sorted(agg.get("SQ_VALU_MFMA_BUSY_CYCLES", [0]))[len(agg.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) // 2] / (1024 * dur_us * 1e-6 * 2.4e9) * 100 if dur else 0
open(os.path.join(P, f"{TAG}_bench_line.json"), "w").write(bench + "\n")
open(os.path.join(P, f"{TAG}_lc_bench_line.json"), "w").write(lc_bench + "\n")
shutil.copy(f("stats_serial/run_kernel_stats.csv"), os.path.join(P, f"{TAG}_bench_kernel_stats.csv"))
shutil.copy(f("stats/run_kernel_stats.csv"), os.path.join(P, f"{TAG}_bench_kernel_stats_overlapped.csv"))
shutil.copy(f("lc_stats/run_kernel_stats.csv"), os.path.join(P, f"{TAG}_lc_kernel_stats.csv"))
json.dump(traffic, open(os.path.join(P, f"{TAG}_traffic.json"), "w"), indent=1)
r = b["roofline"]
t256 = traffic.get("gemm_nt_t256_kernel", {})
doc = f"""# Round {int(TAG[1:3])} — rocprofv3 summaries (MI355X, bf16, batch 16, preset-1 small)

Collected by `tools/profile_round2.sh` (round 3: through `tools/profile_round3.sh`) in one gpurun call (separate rocprofv3 process per pass), assembled by
`tools/write_round2_summary.py` (tables by `tools/summarize_profile.py`, timeline by `tools/step_timeline.py`).  Code state: the
commit this file belongs to.  Un-profiled bench on the same box (`profiles/{TAG}_bench_line.json`): **{b["value"]:.0f} samples/s,
{b["ms_per_step"]:.2f} ms/step**, ELBO of the first step within {b.get("elbo_rel_vs_cpu_port")} (relative) of the CPU restatement; dominant kernel class
`gemm_nt_t256_kernel`: {r["achieved"]:.0f} TFLOP/s = {r["frac"]:.3f} of the 2500 TFLOP/s dense bf16 peak over {r["launches_per_step"]} launches/step
({r["avg_launch_ms"] * 1e3:.0f} µs average, hipEvents).  Box-to-box spread of the bench is about ±5 %; under rocprofv3 the same step reads 5-15 % slower.

## Kernel time per training step, one stream (`SGV_DW_SIDE=0 SGV_LANES=0`; comparable with bench.py's hipEvent numbers)

Command: `SGV_DW_SIDE=0 SGV_LANES=0 rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --cpu-baseline skip`
(raw: `profiles/{TAG}_bench_kernel_stats.csv`).  The 256x256 kernel appears as four template instances (<OUT, MT>: bf16 / fp32-slab
output, one / several taps); its class in bench.py also counts the 128-row tail launch (`gemm_nt_wide64p_kernel`) and the
split-K combine (`t256_reduce_kernel`) of a planned launch.

{serial}
## Kernel time per training step, default mode (weight-gradient side stream + second compute lane)

Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --cpu-baseline skip`
(raw: `profiles/{TAG}_bench_kernel_stats_overlapped.csv`); kernels overlap, so their sum exceeds the step time.

{overl}
Timeline of one step of that run (`tools/step_timeline.py`: BIG = a kernel of >= 100 µs is running, SMALL = only short kernels):

```
{timeline}
```

## HBM-side traffic (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, bench.py --steps 2 --warmup 1)

{fetch}
{write}
Machine-readable: `profiles/{TAG}_traffic.json` (bench.py reads it for `roofline.traffic`).  `gemm_nt_t256_kernel`:
{(t256.get("fetch_bytes", 0) + t256.get("write_bytes", 0)) / 1e9:.2f} GB over {t256.get("launches", 0)} launches per step (its split-K combine included).

## SQ counters of `gemm_nt_t256_kernel` on M=3200, 5120 x 5120, 5 taps (`tests/micro/gemm_bench.py nt256 3200 5120 5120 5`)

Four `--pmc` passes (3 counters each), median over the launches; {dur_us:.0f} µs under the tool (the launch covers rows 0..3071; the
128-row tail is a second kernel):

```
{sq}
```
MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x 2.4 GHz) = {mf:.0f} %.

## Latent conditioner (`bench.py --workload lc`, 512 x 512, batch 16, bf16)

Un-profiled: {lb["value"]:.0f} samples/s, {lb["ms_per_step"]:.2f} ms/step (`profiles/{TAG}_lc_bench_line.json`); dominant GEMM class
{lb["roofline"]["achieved"] if lb.get("roofline") else None} TFLOP/s.  Kernel stats of `rocprofv3 --kernel-trace --stats -- python3 bench.py --workload lc --steps 10 --warmup 3 --cpu-baseline skip`
(raw: `profiles/{TAG}_lc_kernel_stats.csv`; whole run of 15 steps):

{lc_tab}

HBM traffic of its two GEMM operator classes (`lc_gemm_nt`, `lc_gemm_tn` in `profiles/{TAG}_traffic.json`):
{json.dumps({k: v for k, v in traffic.items() if k.startswith("lc_")})}

## Per-layer GEMM table (bench.py --layer-times, hipEvents, one step; top 40 by time)

```
{chr(10).join(layers)}
```
"""
if notes:
    doc += f"\n## Notes\n\n{notes}\n"
open(os.path.join(P, f"{TAG}_summary.md"), "w").write(doc)
print(f"wrote profiles/{TAG}_summary.md:", len(doc.splitlines()), "lines;", f"{b['value']:.0f} samples/s")
