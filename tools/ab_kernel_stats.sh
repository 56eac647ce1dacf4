#!/bin/bash
# Per-kernel A/B on one box: rocprofv3 kernel stats of the single-stream bench, run A with the library named by $1
# (SGV_LIB, e.g. tests/micro/_ab/libsgvae_prev.so) or, if $1 is NAME=VALUE, with that switch in the environment; run B
# with the in-tree library and defaults.  Output: gpurun_out/<tag>/{a,b}.
set -o pipefail
prev=$1; tag=${2:-ab}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SGV_DW_SIDE=0
if [[ "$prev" == *=* ]]; then export "$prev"; else export SGV_LIB=$R/$prev; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -o a -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-baseline skip --no-kernel-timing > $O/a.log 2>&1 || exit 1
if [[ "$prev" == *=* ]]; then unset "${prev%%=*}"; else unset SGV_LIB; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -o b -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-baseline skip --no-kernel-timing > $O/b.log 2>&1 || exit 1
cd $R
python3 - <<PY
import csv, glob
def load(d):
    f = glob.glob(f"$O/{d}/*kernel_stats.csv")[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load("a"), load("b")
names = sorted(set(a) | set(b), key=lambda n: -(a.get(n, (0, 0))[1] + b.get(n, (0, 0))[1]))
ta = sum(v[1] for v in a.values()); tb = sum(v[1] for v in b.values())
print(f"total kernel ms: prev {ta/1e6:.1f}  new {tb/1e6:.1f}")
for n in names[:28]:
    ca, da = a.get(n, (0, 0.0)); cb, db = b.get(n, (0, 0.0))
    print(f"{n[:70]:70s} prev {ca:6d} x {da/max(ca,1)/1e3:8.1f} us   new {cb:6d} x {db/max(cb,1)/1e3:8.1f} us   delta {(db-da)/1e6:+7.2f} ms")
PY
