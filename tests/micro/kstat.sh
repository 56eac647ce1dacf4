#!/bin/bash
# average duration of the kernels whose name contains one of the given substrings over a short profiled bench run:  kstat.sh <tag> substr...
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$1; shift; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o run -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline skip --no-kernel-timing > $O/st.log 2>&1 || exit 1
python3 - "$O/st/run_kernel_stats.csv" "$@" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if any(s in r["Name"] for s in sys.argv[2:]):
        print(f"{float(r['AverageNs'])/1e3:9.1f} us x{r['Calls']:>5s}  {r['Name'][:110]}")
PY
rm -rf $O/st
