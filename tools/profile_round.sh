#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01
# pass 1: kernel trace + stats of the default bench; pass 2/3: FETCH_SIZE / WRITE_SIZE (separate --pmc passes,
# MI355X_MICROARCH.md HBM section); pass 4: SQ counters of the dominant GEMM on its largest shape.
set -o pipefail
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline skip > $O/stats.log 2>&1 || exit 1
echo "[profile] stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline skip --no-kernel-timing > $O/fetch.log 2>&1 || exit 1
echo "[profile] FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline skip --no-kernel-timing > $O/write.log 2>&1 || exit 1
echo "[profile] WRITE_SIZE pass done"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_$n -- python3 $R/tests/micro/gemm_bench.py nt 3200 5120 5120 5 3 > $O/sq_$n.log 2>&1 || exit 1
done
echo "[profile] SQ passes done"
cd $R
tail -n 1 $O/stats.log | cut -c1-400
