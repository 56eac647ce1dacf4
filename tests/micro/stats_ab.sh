#!/bin/bash
# HISTORICAL: the MFMA-statistics build was removed; kept as the record of how it was run.
# GroupNorm statistics of the 256x256 kernel's epilogue through MFMA (in-tree build) vs VALU (SGV_LIB = -DT256_MFMA_STATS=0 build)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-stats_ab}; mkdir -p $O; B=$R/tests/micro/gemm_bench.py
for i in 1 2 3; do
  echo "== valu" >> $O/times.txt; SGV_LIB=$R/tests/micro/_ab/libsgvae_vstats.so STATS=1 python3 $B nt256 3200 95008 1024 1 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt
  echo "== mfma" >> $O/times.txt; STATS=1 python3 $B nt256 3200 95008 1024 1 10 2>&1 | grep -v amdgpu.ids >> $O/times.txt
done
cat $O/times.txt
cd $R; tests/micro/step_ab.sh $1/step 3 "SGV_LIB=$R/tests/micro/_ab/libsgvae_vstats.so" "-"
