#!/bin/bash
# rocprofv3 evidence for one round on the GPU box (through gpurun, from the repo root):  tools/profile_round2.sh r02
#   stats        kernel trace + stats of the default bench (two streams + second lane)           -> <tag>/stats
#   stats_serial same with everything on one stream (comparable with bench.py's hipEvent timing)  -> <tag>/stats_serial
#   fetch/write  FETCH_SIZE / WRITE_SIZE, separate --pmc passes (MI355X_MICROARCH.md HBM section) -> <tag>/fetch, <tag>/write
#   sq_*         SQ counters of the 256x256 kernel on M=3200, 5120x5120, 5 taps                    -> <tag>/sq_*
#   lc_*         the latent-conditioner bench: stats + FETCH_SIZE / WRITE_SIZE                     -> <tag>/lc_stats, lc_fetch, lc_write
set -o pipefail
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
python3 $R/bench.py --steps 30 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err || exit 1
python3 $R/bench.py --steps 5 --warmup 3 --cpu-baseline skip --layer-times > /dev/null 2> $O/layers.log || exit 1
python3 $R/bench.py --workload lc --steps 20 --warmup 5 > $O/lc_bench_line.json 2> $O/lc_bench_line.err || exit 1
echo "[profile] plain bench lines done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline skip > $O/stats.log 2>&1 || exit 1
SGV_DW_SIDE=0 SGV_LANES=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -o run -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-baseline skip > $O/stats_serial.log 2>&1 || exit 1
echo "[profile] stats passes done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline skip --no-kernel-timing > $O/fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-baseline skip --no-kernel-timing > $O/write.log 2>&1 || exit 1
echo "[profile] FETCH_SIZE / WRITE_SIZE passes done"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  n=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sq_$n -o run -- python3 $R/tests/micro/gemm_bench.py nt256 3200 5120 5120 5 3 > $O/sq_$n.log 2>&1 || exit 1
done
echo "[profile] SQ passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lc_stats -o run -- python3 $R/bench.py --workload lc --steps 10 --warmup 3 --cpu-baseline skip > $O/lc_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/lc_fetch -o run -- python3 $R/bench.py --workload lc --steps 2 --warmup 1 --cpu-baseline skip > $O/lc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/lc_write -o run -- python3 $R/bench.py --workload lc --steps 2 --warmup 1 --cpu-baseline skip > $O/lc_write.log 2>&1 || exit 1
echo "[profile] latent-conditioner passes done"
cd $R
find $O -name "*.csv" -size +40M -delete     # the merge limit is 64 MiB: traces of the long passes are summarised on the box instead
tail -n 1 $O/bench_line.json | cut -c1-300
