#!/bin/bash
# rocprofv3 kernel stats of the bench (one stream) -> gpurun_out/<tag>/stats.txt
tag=${1:-prof}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SGV_DW_SIDE=${SGV_DW_SIDE:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline skip --no-kernel-timing > $O/bench.log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = glob.glob("$O/p/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (13 steps + setup):", tot/1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(f'{r["Name"][:90]:90s} {int(r["Calls"]):6d} x {float(r["AverageNs"])/1e3:9.1f} us = {float(r["TotalDurationNs"])/13e6:7.3f} ms/step')
PY
tail -2 $O/bench.log | cut -c1-200
