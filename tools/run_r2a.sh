set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > gpurun_out/r2a/test.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/test.txt
tail -15 gpurun_out/r2a/test.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --cpu-baseline skip --layer-times > gpurun_out/r2a/bench.txt 2> gpurun_out/r2a/bench_err.txt; echo "bench rc=$?"
cat gpurun_out/r2a/bench.txt; grep -v "^\[bench\] init" gpurun_out/r2a/bench_err.txt | head -70
