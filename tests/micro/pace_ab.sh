#!/bin/bash
# HISTORICAL: SGV_PACE / SGV_SLAB_DEFER were removed together with the experiment; kept as the record of how it was run.
# Round 3 A/B: paced AdamW (SGV_PACE) and deferred split-K combines (SGV_SLAB_DEFER) on the whole step, alternating on one box.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-pace_ab}
mkdir -p $O
cd $R
one() { name=$1; shift; env "$@" python3 bench.py --steps 40 --warmup 8 --cpu-baseline skip --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])" >> $O/ab.txt; }
for i in 1 2; do
  one base SGV_PACE=0 SGV_SLAB_DEFER=0
  one pace22 SGV_PACE=1 SGV_SLAB_DEFER=0
  one pace22_defer SGV_PACE=1 SGV_SLAB_DEFER=1
  one defer SGV_PACE=0 SGV_SLAB_DEFER=1
  one pace12 SGV_PACE=1 SGV_PACE_TILES_PER_GF=12
  one pace35 SGV_PACE=1 SGV_PACE_TILES_PER_GF=35
  one pace22_min30 SGV_PACE=1 SGV_PACE_MIN_GF=30
done
cat $O/ab.txt
