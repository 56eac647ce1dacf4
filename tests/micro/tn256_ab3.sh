#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-tn256_ab3}; mkdir -p $O; B=$R/tests/micro/gemm_bench.py
for shape in "3200 1024 95008 1" "3200 95008 1024 1" "3200 5120 5120 5"; do
  for ord in 0 2; do
    echo "== order $ord tn $shape" >> $O/times.txt
    SGV_TN256_ORDER=$ord USE_TR=4 python3 $B tn $shape 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt
  done
done
cat $O/times.txt
