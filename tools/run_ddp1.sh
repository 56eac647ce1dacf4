#!/bin/bash
# forced one-rank data-parallel run (callback path and native path) next to the plain step
O=gpurun_out/ddp1; mkdir -p $O
python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/plain.txt 2>&1; echo "plain $(grep -o '"ms_per_step": [0-9.]*' $O/plain.txt)"
SGV_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/ddp.txt 2>&1; echo "forced ddp (torch.distributed) $(grep -o '"ms_per_step": [0-9.]*' $O/ddp.txt)"
SGV_FORCE_DDP=1 SGV_DDP_NATIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/native.txt 2>&1; echo "forced ddp (native RCCL) $(grep -o '"ms_per_step": [0-9.]*' $O/native.txt)"
python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/plain2.txt 2>&1; echo "plain $(grep -o '"ms_per_step": [0-9.]*' $O/plain2.txt)"
# the same two with every bucket really going through RCCL (one-rank ncclAllReduce: pack -> all-reduce -> unpack; what N > 1 runs)
SGV_FORCE_DDP=1 SGV_FORCE_COLLECTIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/ddp_coll.txt 2>&1; echo "forced collectives (torch.distributed) $(grep -o '"ms_per_step": [0-9.]*' $O/ddp_coll.txt)"
SGV_FORCE_DDP=1 SGV_FORCE_COLLECTIVE=1 SGV_DDP_NATIVE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29614 bench.py --gpus 1 --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/native_coll.txt 2>&1; echo "forced collectives (native RCCL) $(grep -o '"ms_per_step": [0-9.]*' $O/native_coll.txt)"
