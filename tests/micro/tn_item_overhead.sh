#!/bin/bash
# per-item overhead of the 256x256 TN kernel: the same widths at M = 3200 and M = 12800 (50 / 200 K-tiles per item)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-tnov}; mkdir -p $O; cd $R
for shape in "tn 3200 5120 5120 5" "tn 12800 5120 5120 5" "tn 3200 1024 95008 1" "tn 12800 1024 95008 1" "nt256 3200 5120 5120 5" "nt256 3072 5120 5120 5" "tn 3200 5120 5120 1" "tn 12800 5120 5120 1"; do
  USE_TR=4 python3 tests/micro/gemm_bench.py $shape 5 2>/dev/null | tail -1 >> $O/ov.txt
done
cat $O/ov.txt
