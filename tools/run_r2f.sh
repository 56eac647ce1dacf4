#!/bin/bash
O=gpurun_out/r2f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_engine_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt
tail -8 $O/test.txt
for c in 3 1 2 4 3 1; do
SGV_DW_CHUNKS=$c python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/bench_c$c.txt 2>&1
echo "chunks=$c $(grep -o '"ms_per_step": [0-9.]*' $O/bench_c$c.txt)"
done
