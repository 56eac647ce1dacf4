#!/bin/bash
# latent-conditioner bench A/B on one box: im2col lowering vs implicit-GEMM convolutions, then the conditioner test files
O=gpurun_out/${1:-lcab}; mkdir -p $O
SGV_LC_IMPLICIT=0 python3 bench.py --workload lc --steps 20 --warmup 5 --cpu-baseline skip > $O/lc_im2col.json 2> $O/lc_im2col.err || exit 1
python3 bench.py --workload lc --steps 20 --warmup 5 --cpu-baseline skip > $O/lc_implicit.json 2> $O/lc_implicit.err || exit 1
cut -c1-200 $O/lc_im2col.json; cut -c1-200 $O/lc_implicit.json
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_lc_loop_gpu.py tests/test_configs_gpu.py -x -q -m gpu > $O/test.txt 2>&1; tail -5 $O/test.txt
