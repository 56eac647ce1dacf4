#!/bin/bash
# bench with the library of the previous commit (tests/micro/ab/libsgvae_prev.so, built by hand) and the current one, interleaved:
#   tools/run_ab_lib.sh <tag> [pairs] [extra bench.py arguments, e.g. --workload lc]
O=gpurun_out/${1:-ablib}; mkdir -p $O; N=${2:-2}; shift; shift
for i in $(seq 1 $N); do
  SGV_LIB=$PWD/tests/micro/ab/libsgvae_prev.so python3 bench.py --steps 40 --warmup 5 --cpu-baseline skip "$@" > $O/prev$i.json 2> $O/prev$i.err || exit 1
  python3 bench.py --steps 40 --warmup 5 --cpu-baseline skip "$@" > $O/cur$i.json 2> $O/cur$i.err || exit 1
done
for i in $(seq 1 $N); do for f in prev$i cur$i; do echo -n "$f "; python3 -c "import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done; done
