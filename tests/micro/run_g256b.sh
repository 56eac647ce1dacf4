#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/g256; mkdir -p $O
B=$R/tests/micro/_ab
echo "== small (correctness)"; timeout -k 10 100 $B/g256 reps=2 600 512 200 3 1  520 264 520 1 1  600 512 520 3 2  1000 768 1024 5 1  1000 768 1032 5 1 800 1024 4104 1 3 2>&1 | grep -v "\.\.\.$"
echo "== addend"; timeout -k 10 100 $B/g256 reps=2 addend=1 1000 768 1024 5 1 1000 768 1024 1 2 2>&1 | grep -v "\.\.\.$"
for b in g256 g256_st0; do
  echo "== $b"; timeout -k 10 100 $B/$b reps=5 check=1 old=0 3072 5120 5120 5 1  3200 5120 5120 5 1 3072 5120 1024 1 1  3200 95008 1024 1 1 3200 1024 95008 1 0 2>&1 | grep -v "\.\.\.$" | tail -6
done
echo "== stats"; timeout -k 10 200 $B/g256 reps=2 stats=1 old=0 1000 1024 520 1 1  3200 95008 1024 1 1 2>&1 | grep -v "\.\.\.$"
