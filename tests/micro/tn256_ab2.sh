#!/bin/bash
# fine-grained vs coarse LDS waits of gemm_tn_t256 (SGV_LIB = the coarse build), micro shapes + whole step
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-tn256_ab2}
mkdir -p $O
B=$R/tests/micro/gemm_bench.py
for shape in "3200 95008 1024 1" "3200 1024 95008 1" "3200 5120 5120 5" "3200 2560 2560 5"; do
  echo "== coarse tn $shape" >> $O/times.txt
  SGV_LIB=$R/tests/micro/_ab/libsgvae_nofw.so USE_TR=4 python3 $B tn $shape 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt
  echo "== fine tn $shape" >> $O/times.txt
  USE_TR=4 python3 $B tn $shape 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt
done
cat $O/times.txt
cd $R
tests/micro/step_ab.sh $1/step 2 "SGV_LIB=$R/tests/micro/_ab/libsgvae_nofw.so" "-"
