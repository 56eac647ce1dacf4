#!/bin/bash
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_bigfix_gpu.py -x -q -m gpu > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt
tail -6 $O/test.txt
for f in 1 0 1 0; do
SGV_SPLITK_FUSED=$f python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/vae_$f.txt 2>&1; echo "fused=$f $(grep -o '"ms_per_step": [0-9.]*' $O/vae_$f.txt)"
done
