#!/bin/bash
# stream priority levels of the auxiliary streams (lane / side / opt): plain single-GPU step and the one-GPU rehearsal of the
# data-parallel step, two rounds each; then the stream -> hardware-queue map of the default configuration from a kernel trace
O=gpurun_out/prio; mkdir -p $O
plain() { env "$@" python bench.py --steps 40 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
ddp() { env "$@" SGV_FORCE_DDP=1 SGV_FORCE_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2; }
for cfg in "0 0 0" "-1 0 0" "-1 1 1" "0 1 1" "-1 -1 1" "-1 1 0"; do
  set -- $cfg
  E="SGV_PRIO_LANE=$1 SGV_PRIO_SIDE=$2 SGV_PRIO_OPT=$3 SGV_PRIO_WIRE=$3"
  echo "lane=$1 side=$2 opt=$3: plain $(plain $E) $(plain $E)  ddp-torch $(ddp $E SGV_DDP_WIRE=0)  ddp-native $(ddp $E SGV_DDP_NATIVE=1)"
done | tee $O/ab.txt
