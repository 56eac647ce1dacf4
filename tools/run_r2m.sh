#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r2m; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
SGV_DW_SIDE=0 SGV_LANES=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 $R/bench.py --steps 6 --warmup 2 --cpu-baseline skip --no-kernel-timing > $O/bench.log 2>&1
cd $R; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/p/p_kernel_stats.csv")))
for r in rows:
    if "conv_gn" in r["Name"] or "gn_fwd_fused" in r["Name"] or "gemm_nt_kernel" in r["Name"] or "reduce4" in r["Name"] or "wide64p" in r["Name"]:
        print(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6/8)
PY
