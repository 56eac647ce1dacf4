#!/bin/bash
O=gpurun_out/r2k; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py tests/test_lc_loop_gpu.py tests/test_e2e_gpu.py tests/test_modules_gpu.py -x -q -m gpu > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt
tail -15 $O/test.txt
python bench.py --workload lc --steps 20 --warmup 5 --cpu-baseline skip > $O/lc.txt 2>&1; echo "lc $(grep -o '"value": [0-9.]*' $O/lc.txt | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/lc.txt | head -1)"
