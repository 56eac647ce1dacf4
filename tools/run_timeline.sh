#!/bin/bash
# kernel trace of the default bench (two streams + lanes) -> step timeline; extra arguments go to bench.py (e.g. --batch 8)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/tl; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/bench.py --steps 8 --warmup 3 --cpu-baseline skip --no-kernel-timing "$@" > $O/bench.log 2>&1
cd $R; python3 tools/step_timeline.py $O/p/p_kernel_trace.csv
