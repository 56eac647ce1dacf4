#!/bin/bash
# Round 3 A/B of the work-item orders (gemm_tn_w2 persistent patches, gemm_nt_t256 row-tile bands + non-temporal weights):
# un-profiled time (hipEvents) and FETCH_SIZE per launch for the four big weight-gradient shapes and the recon head.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-order_ab}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/tests/micro/gemm_bench.py
run() {   # name, env..., -- args
  name=$1; shift
  envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "== $name: ${envs[*]} $*" >> $O/times.txt
  env "${envs[@]}" DUMMY=1 python3 $B "$@" 10 >> $O/times.txt 2>&1 || return 1
  ( export "${envs[@]}" DUMMY=1; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/$name -o run -- python3 $B "$@" 2 > $O/$name.log 2>&1 ) || return 1
  echo "== $name: ${envs[*]} $*" >> $O/fetch.txt
  python3 $R/tests/micro/pmc_by_kernel.py $O/$name gemm_tn gemm_nt t256 sum_slabs >> $O/fetch.txt
  rm -rf $O/$name
}
for P in 0 1; do
  run tn_recon_p$P SGV_TN_PERSIST=$P -- tn 3200 95008 1024 1 || exit 1
  run tn_enc0_p$P SGV_TN_PERSIST=$P -- tn 3200 1024 95008 1 || exit 1
  run tn_k5_p$P SGV_TN_PERSIST=$P -- tn 3200 5120 5120 5 || exit 1
  run tn_k5m_p$P SGV_TN_PERSIST=$P -- tn 3200 2560 2560 5 || exit 1
done
run tn_k5_62 SGV_TN_PT1=6 SGV_TN_PT2=2 -- tn 3200 5120 5120 5 || exit 1
run tn_k5_121 SGV_TN_PT1=12 SGV_TN_PT2=1 -- tn 3200 5120 5120 5 || exit 1
run tn_recon_84 SGV_TN_PT1=8 SGV_TN_PT2=4 -- tn 3200 95008 1024 1 || exit 1
run tn_recon_322 SGV_TN_PT1=32 SGV_TN_PT2=2 -- tn 3200 95008 1024 1 || exit 1
run tn_enc0_416 SGV_TN_PT1=4 SGV_TN_PT2=16 -- tn 3200 1024 95008 1 || exit 1
echo "[order_ab] tn done"
run nt_recon_base STATS=1 SGV_T256_BAND=0 SGV_T256_WNT=0 -- nt256 3200 95008 1024 1 || exit 1
run nt_recon_auto STATS=1 -- nt256 3200 95008 1024 1 || exit 1
run nt_recon_b5 STATS=1 SGV_T256_BAND=5 SGV_T256_WNT=0 -- nt256 3200 95008 1024 1 || exit 1
run nt_recon_b7n STATS=1 SGV_T256_BAND=7 SGV_T256_WNT=1 -- nt256 3200 95008 1024 1 || exit 1
run nt_recon_b7 STATS=1 SGV_T256_BAND=7 SGV_T256_WNT=0 -- nt256 3200 95008 1024 1 || exit 1
run nt_recon_b4 STATS=1 SGV_T256_BAND=4 SGV_T256_WNT=0 -- nt256 3200 95008 1024 1 || exit 1
run nt_recon_b0n STATS=1 SGV_T256_BAND=0 SGV_T256_WNT=1 -- nt256 3200 95008 1024 1 || exit 1
echo "[order_ab] nt done"
cat $O/times.txt
cat $O/fetch.txt
