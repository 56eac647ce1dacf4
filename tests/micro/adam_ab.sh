#!/bin/bash
# AdamW cache-policy A/B on one box, alternating:  tests/micro/adam_ab.sh <tag> <pairs> "ENV=.." ...   ("-" = no env)
# prints per spec: samples/s, ms/step, the optimizer pass timed alone (roofline_adamw.ms_alone)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-adam_ab}; N=${2:-2}; shift; shift
mkdir -p $O; cd $R
for i in $(seq 1 $N); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then envs=(DUMMY=1); else read -r -a envs <<< "$spec"; fi
    env "${envs[@]}" python3 bench.py --steps 40 --warmup 8 --cpu-baseline skip 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', d['value'], d['ms_per_step'], d['roofline_adamw']['ms_alone'], d['roofline']['frac'], d['roofline_gemm_tn']['frac'])" >> $O/ab.txt
  done
done
cat $O/ab.txt
