#!/bin/bash
# kernel stats of the latent-conditioner bench -> gpurun_out/<tag>/lc_stats
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-lcstats}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lc_stats -o run -- python3 $R/bench.py --workload lc --steps 10 --warmup 3 --cpu-baseline skip > $O/lc_stats.log 2>&1 || exit 1
cd $R; find $O -name "*kernel_trace.csv" -size +40M -delete
head -40 $O/lc_stats/run_kernel_stats.csv | cut -c1-160
