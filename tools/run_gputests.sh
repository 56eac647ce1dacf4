#!/bin/bash
# full GPU test tier (what the driver runs at round end) + smoke; output under gpurun_out/<tag>
tag=${1:-tests}
mkdir -p gpurun_out/$tag
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/$tag/test.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/$tag/test.txt
tail -12 gpurun_out/$tag/test.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/$tag/smoke.txt
