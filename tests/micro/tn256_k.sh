#!/bin/bash
# does the 256x256 weight-gradient kernel's rate depend on the K-tiles per item?  one round of 256 items, M = 3200 .. 25600
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-tn256_k}; mkdir -p $O; B=$R/tests/micro/gemm_bench.py
for M in 3200 6400 12800 25600; do
  USE_TR=4 python3 $B tn $M 4096 4096 1 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt
  python3 $B nt256 4096 4096 $M 1 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt
done
cat $O/times.txt
