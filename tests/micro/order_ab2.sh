#!/bin/bash
# Round 3, second A/B: stream policies of the banded recon-head launch (time + FETCH_SIZE), then the whole step with / without bands.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${1:-order_ab2}
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
for S in 0 1 2 3; do
  run nt_recon_b7s$S STATS=1 SGV_T256_BAND=7 SGV_T256_STRM=$S -- nt256 3200 95008 1024 1 || exit 1
done
run nt_recon_b0s2 STATS=1 SGV_T256_BAND=0 SGV_T256_STRM=2 -- nt256 3200 95008 1024 1 || exit 1
run nt_recon_b0s0 STATS=1 SGV_T256_BAND=0 SGV_T256_STRM=0 -- nt256 3200 95008 1024 1 || exit 1
echo "[ab2] recon variants done"
cd $R
for i in 1 2; do
  python3 bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | cut -c1-220 >> $O/bench_default.txt || exit 1
  SGV_T256_BAND=0 SGV_TN_PERSIST=0 python3 bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | cut -c1-220 >> $O/bench_r2order.txt || exit 1
done
SGV_T256_STRM=2 python3 bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | cut -c1-220 >> $O/bench_strm2.txt || exit 1
cat $O/times.txt $O/fetch.txt
echo default; cat $O/bench_default.txt; echo r2order; cat $O/bench_r2order.txt; echo strm2; cat $O/bench_strm2.txt
