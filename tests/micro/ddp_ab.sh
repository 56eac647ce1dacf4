#!/bin/bash
# one-GPU rehearsal of the data-parallel step (every bucket through a one-rank RCCL all-reduce), alternating env specs:
#   tests/micro/ddp_ab.sh <tag> <rounds> "ENV=.." ...   ("-" = no env)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-ddp_ab}; N=${2:-2}; shift; shift
mkdir -p $O; cd $R
port=29700
for i in $(seq 1 $N); do
  for spec in "$@"; do
    if [ "$spec" = "-" ]; then envs=(DUMMY=1); else read -r -a envs <<< "$spec"; fi
    port=$((port + 1))
    env "${envs[@]}" SGV_FORCE_DDP=1 SGV_FORCE_COLLECTIVE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus 1 --steps 40 --warmup 8 --cpu-baseline skip --no-kernel-timing 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', d['value'], d['ms_per_step'], d['config'].get('ddp_path'), d['config'].get('aux_streams_overlap'))" >> $O/ab.txt
  done
done
cat $O/ab.txt
