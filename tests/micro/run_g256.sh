#!/bin/bash
# gpurun helper: runs the GEMM harness on a list of shapes; output under gpurun_out/g256/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/g256; mkdir -p $O
B=$R/tests/micro/_ab/${BIN:-g256}
timeout -k 10 120 $B reps=3 600 512 200 3 1  520 264 520 1 1  600 512 520 3 2  1000 768 1024 5 1 > $O/small.txt 2>&1 || { echo "small failed"; cat $O/small.txt; exit 1; }
cat $O/small.txt
timeout -k 10 300 $B reps=5 3072 5120 5120 5 1  3200 5120 5120 5 1  3200 1024 95008 1 0  3200 5120 1024 1 1  3200 1024 5120 1 0  3200 2560 2560 5 0 3200 95008 1024 1 1 > $O/big.txt 2>&1 || { echo "big failed"; cat $O/big.txt; exit 1; }
cat $O/big.txt
timeout -k 10 200 $B reps=3 stats=1 old=0 T=200 1000 1024 520 1 1  3200 95008 1024 1 1 > $O/stats.txt 2>&1 || { echo "stats failed"; cat $O/stats.txt; exit 1; }
cat $O/stats.txt
