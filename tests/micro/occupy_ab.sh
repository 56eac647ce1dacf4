#!/bin/bash
# How do the persistent GEMMs cope when part of the chip is occupied by another stream's resident workgroups (a collective's channels)?
#   tests/micro/occupy_ab.sh <tag>      prints best-of-5 times: alone, then with 16 / 32 / 64 spinning workgroups of 512 threads
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-occupy}; mkdir -p $O; cd $R
# (weight-gradient shapes only: the nt256 test hook allocates its workspaces, which waits for the spinning workgroups)
for shape in "tn 3200 5120 5120 5" "tn 3200 1024 95008 1" "tn 3200 95008 1024 1" "tn 3200 2560 2560 5"; do
  for occ in "" "32,512,0" "16,512,65536" "32,512,65536" "64,512,65536"; do
    echo "== occupy [$occ] $shape" >> $O/occ.txt
    OCCUPY=$occ USE_TR=4 python3 tests/micro/gemm_bench.py $shape 5 2>/dev/null | tail -1 >> $O/occ.txt
    echo "== occupy [$occ] work-stealing $shape" >> $O/occ.txt
    OCCUPY=$occ USE_TR=7 python3 tests/micro/gemm_bench.py $shape 5 2>/dev/null | tail -1 >> $O/occ.txt
  done
done
cat $O/occ.txt
