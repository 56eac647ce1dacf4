#!/bin/bash
# Round 3 evidence (through gpurun, from the repo root):  tools/profile_round3.sh [tag]
# = tools/profile_round2.sh (bench lines, kernel stats one-stream / overlapped, FETCH_SIZE / WRITE_SIZE passes, SQ counters of
# the 256x256 kernel, conditioner passes) + the `--size large` line with and without the activation-recompute timing hook.
set -o pipefail
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
$R/tools/profile_round2.sh $tag || exit 1
cd $R
for i in 1 2; do
  python3 bench.py --size large --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | tail -n 1 >> $O/large_plain.json || exit 1
  python3 bench.py --size large --recompute --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | tail -n 1 >> $O/large_recompute.json || exit 1
done
python3 - <<PY
import json
for f in ("large_plain", "large_recompute"):
    for l in open("$O/" + f + ".json"):
        d = json.loads(l); c = d["config"]
        print(f, d["value"], d["ms_per_step"], c.get("resident_gib"), c.get("recompute_gib"))
PY
