#!/bin/bash
# lanes on/off: parity subset + bench
O=gpurun_out/r2d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py tests/test_engine_gpu.py tests/test_bigfix_gpu.py -x -q -m gpu > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt
tail -5 $O/test.txt
for l in 1 0 1 0; do
SGV_LANES=$l python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/bench_l$l.txt 2>&1
echo "lanes=$l $(grep -o '"ms_per_step": [0-9.]*' $O/bench_l$l.txt)"
done
