#!/bin/bash
# latent-conditioner bench A/B on one box: im2col lowering and separate tail passes vs the default (implicit-GEMM convolutions,
# direct stem, fused block tail), then the conditioner tests
O=gpurun_out/${1:-lcab}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_lc_loop_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/test.txt 2>&1; tail -5 $O/test.txt
grep -q " passed" $O/test.txt || exit 1
SGV_LC_IMPLICIT=0 SGV_LC_FUSED_TAIL=0 python3 bench.py --workload lc --steps 20 --warmup 5 --cpu-baseline skip > $O/lc_im2col.json 2> $O/lc_im2col.err || exit 1
python3 bench.py --workload lc --steps 20 --warmup 5 --cpu-baseline skip > $O/lc_default.json 2> $O/lc_default.err || exit 1
for f in lc_im2col lc_default; do echo -n "$f "; cut -c95-200 $O/$f.json; done
