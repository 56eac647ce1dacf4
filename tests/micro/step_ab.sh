#!/bin/bash
# whole-step A/B on one box, alternating:  tests/micro/step_ab.sh <tag> <pairs> "ENV_A=.. ENV_B=.." "ENV_C=.."   (first spec = baseline; "-" = no env)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-step_ab}; N=${2:-2}; shift; shift
mkdir -p $O; cd $R
for i in $(seq 1 $N); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then envs=(DUMMY=1); else read -r -a envs <<< "$spec"; fi
    env "${envs[@]}" python3 bench.py --steps 40 --warmup 8 --cpu-baseline skip --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
cat $O/ab.txt
