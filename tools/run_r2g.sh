#!/bin/bash
O=gpurun_out/r2g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py tests/test_lc_loop_gpu.py tests/test_e2e_gpu.py tests/test_fullsize_gpu.py tests/test_bigfix_gpu.py -x -q -m gpu > $O/test.txt 2>&1; echo "pytest rc=$?" >> $O/test.txt
tail -6 $O/test.txt
python bench.py --workload lc --steps 20 --warmup 5 --cpu-baseline skip > $O/lc.txt 2>&1; echo "lc $(grep -o '"value": [0-9.]*' $O/lc.txt | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/lc.txt | head -1)"
python bench.py --steps 30 --warmup 5 --cpu-baseline skip --no-kernel-timing > $O/vae.txt 2>&1; echo "vae $(grep -o '"ms_per_step": [0-9.]*' $O/vae.txt)"
