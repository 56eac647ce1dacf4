#!/bin/bash
# 256 x 256 weight-gradient kernel (gemm256tn.hip) against gemm_tn_w2 on the big shapes: hipEvent time and FETCH_SIZE per launch.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-tn256_ab}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/micro/gemm_bench.py
for shape in "3200 95008 1024 1" "3200 1024 95008 1" "3200 5120 5120 5" "3200 2560 2560 5" "3200 1024 5120 1" "3200 5120 1024 1"; do
  for U in 5 4; do
    echo "== USE_TR=$U tn $shape" >> $O/times.txt
    USE_TR=$U python3 $B tn $shape 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt || exit 1
  done
done
for shape in "3200 95008 1024 1" "3200 5120 5120 5"; do
  for U in 5 4; do
    ( export USE_TR=$U; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f_$U -o run -- python3 $B tn $shape 2 > $O/f.log 2>&1 ) || exit 1
    echo "== USE_TR=$U tn $shape" >> $O/fetch.txt
    python3 $R/tests/micro/pmc_by_kernel.py $O/f_$U gemm_tn >> $O/fetch.txt
    rm -rf $O/f_$U
  done
done
cat $O/times.txt $O/fetch.txt
