#!/bin/bash
# FETCH_SIZE / duration of one GEMM shape: tests/micro/pmc_fetch.sh <tag> <gemm_bench args...>   (env SPLITK passes through)
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/tests/micro/gemm_bench.py "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
F=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -n 1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "gemm" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][:50]].append((float(r["Counter_Value"]) * 1024 * 2, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in agg.items():
    b = sorted(x[0] for x in v)[len(v) // 2]; d = sorted(x[1] for x in v)[len(v) // 2]
    print(f"{k}: launches {len(v)} median fetch {b/1e9:.3f} GB  median dur {d/1e3:.0f} us")
PY
