#!/bin/bash
# 128 x 512 tiles of gemm_nt_t256 (no tail launch) against 256 x 256 + tail: per-layer hipEvent times and the whole step
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/${1:-wide_ab}; mkdir -p $O; cd $R
for W in 0 2; do
  SGV_T256_WIDE=$W python3 bench.py --steps 3 --warmup 2 --cpu-baseline skip --layer-times > /dev/null 2> $O/layers_w$W.log
  grep "gemm_nt_t256" $O/layers_w$W.log | sort -k6 | awk '{print $1,$2,$3,$4,$6,$7,$8,$9,$10,$11}' > $O/t256_w$W.txt
done
paste -d'|' $O/t256_w0.txt $O/t256_w2.txt | cut -c1-230
tests/micro/step_ab.sh $1/step 2 "SGV_T256_WIDE=0" "SGV_T256_WIDE=1" "SGV_T256_WIDE=2"
