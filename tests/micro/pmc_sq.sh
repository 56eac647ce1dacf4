#!/bin/bash
# SQ counters of the GEMM kernels of one gemm_bench shape: tests/micro/pmc_sq.sh <tag> <gemm_bench args...>
R=${GRAFT_REPO_ROOT:-$PWD}
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU"; do
  n=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq_${tag}_$n -- python3 $R/tests/micro/gemm_bench.py "$@" > $R/gpurun_out/sq_${tag}_$n.log 2>&1
done
cd $R
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/sq_{sys.argv[1]}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = sorted(v); print(f"{k[0]:42s} {k[1]:28s} {v[len(v)//2]:.4e}")
PY
