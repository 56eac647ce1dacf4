#!/bin/bash
# one-GPU rehearsal of the data-parallel step (SGV_FORCE_COLLECTIVE=1): hardware-queue count x wire stream x issue path
O=gpurun_out/ddpab; mkdir -p $O
run() {  # name, env...
  local name=$1; shift
  env "$@" SGV_FORCE_DDP=1 SGV_FORCE_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29620 \
      bench.py --gpus 1 --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/$name.txt 2>&1
  echo "$name $(grep -o '"ms_per_step": [0-9.]*' $O/$name.txt)"
}
for q in 4 8; do
  env GPU_MAX_HW_QUEUES=$q python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/plain_q$q.txt 2>&1; echo "plain q$q $(grep -o '"ms_per_step": [0-9.]*' $O/plain_q$q.txt)"
  for w in 0 1; do
    run torch_q${q}_wire$w GPU_MAX_HW_QUEUES=$q SGV_DDP_WIRE=$w
  done
  run native_q$q GPU_MAX_HW_QUEUES=$q SGV_DDP_NATIVE=1
  run torch_q${q}_wire0_noearly GPU_MAX_HW_QUEUES=$q SGV_DDP_WIRE=0 SGV_DDP_EARLY=0
done
