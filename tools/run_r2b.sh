timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -4
bash tools/prof_bench.sh p2 2>&1 | head -42
grep -o '"value": [0-9.]*, "unit": "samples/s", "n_gpus": 1, "steps": 10, "warmup": 3, "ms_per_step": [0-9.]*' gpurun_out/p2/bench.log
python bench.py --steps 20 --warmup 5 --cpu-baseline skip --no-kernel-timing 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "samples/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": [0-9.]*'
