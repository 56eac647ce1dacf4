timeout -k 10 700 python -m pytest tests/test_configs_gpu.py tests/test_ops_gpu.py tests/test_e2e_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "--- lc bench (planner)"; timeout -k 10 300 python bench.py --workload lc --image 512 --steps 10 --warmup 3 --cpu-baseline skip 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'] and (d['roofline']['achieved'], d['roofline']['ms_per_step'], d['roofline']['launches_per_step']), {k:(d[k]['achieved'], d[k]['ms_per_step']) for k in d if k.startswith('roofline_')})"
echo "--- lc bench (library comparator)"; SGV_VENDOR_GEMM=1 timeout -k 10 300 python bench.py --workload lc --image 512 --steps 10 --warmup 3 --cpu-baseline skip 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
