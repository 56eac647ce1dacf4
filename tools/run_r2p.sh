#!/bin/bash
O=gpurun_out/r2p; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_bigfix_gpu.py tests/test_loop_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt
tail -6 $O/test.txt
for f in 1 2 3; do
python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/vae_$f.txt 2>&1; echo "run $f $(grep -o '"ms_per_step": [0-9.]*' $O/vae_$f.txt)"
done
