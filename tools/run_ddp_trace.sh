#!/bin/bash
# kernel trace of the one-GPU rehearsal of the data-parallel step (every bucket through RCCL); prints the kernels >= 40 us of one step
# with their queue, start and duration.  Extra environment (e.g. SGV_DDP_NATIVE=1, SGV_DDP_EARLY=0) is inherited.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/ddptl${1:+_$1}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29650 SGV_FORCE_DDP=1 SGV_FORCE_COLLECTIVE=1
rocprofv3 --kernel-trace --output-format csv -d $O/p -o p -- python3 $R/bench.py --steps 6 --warmup 3 --cpu-baseline skip --no-kernel-timing > $O/bench.log 2>&1
cd $R; grep -o '"ms_per_step": [0-9.]*' $O/bench.log
python3 - $O/p/p_kernel_trace.csv > $O/step.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sn_wt_u_kernel" in r["Kernel_Name"]]
lo, hi = marks[len(marks) // 2], marks[len(marks) // 2 + 1]
t0 = int(rows[lo]["Start_Timestamp"])
qs = {}
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = qs.setdefault(r.get("Queue_Id", "?"), len(qs))
    if e - s >= 40000:
        print(f"q{q} {(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
print(f"step {(int(rows[hi]['Start_Timestamp']) - t0) / 1e6:.3f} ms, {hi - lo} kernels")
PY
tail -1 $O/step.txt
