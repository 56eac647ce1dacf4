"""HBM throughput of the input-pipeline kernels at the full sample shape.
   python tests/micro/pipeline_bench.py [params_per_chunk]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simulgen_vae_amd
from simulgen_vae_amd import engine as E
lib = E.load_library()
c = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, N = 200, 95008
R = c * T
src = torch.rand((R, N), device="cuda") * 3 - 1
mn = torch.empty(N, device="cuda"); mx = torch.empty(N, device="cuda"); sc = torch.empty(N, device="cuda"); of = torch.empty(N, device="cuda")
dst = torch.empty(R * N, dtype=torch.bfloat16, device="cuda")
vp = lambda t: C.c_void_p(t.data_ptr())
assert lib.sgv_minmax_fit(vp(src), R, N, vp(mn), vp(mx), 0, None) == 0
assert lib.sgv_minmax_coeffs(vp(mn), vp(mx), N, -0.7, 0.7, vp(sc), vp(of), None) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(5):
    e0.record(); assert lib.sgv_scale_convert(1, vp(src), vp(sc), vp(of), vp(dst), R, N, None) == 0; e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
gb = R * N * 6 / 1e9
print(f"scale_convert {R}x{N} fp32->bf16: {best*1e3:.0f} us, {gb/best*1e3/1e3:.2f} TB/s ({gb:.2f} GB algorithmic: 4 B read + 2 B written per value)")
best = 1e9
for _ in range(5):
    e0.record(); assert lib.sgv_minmax_fit(vp(src), R, N, vp(mn), vp(mx), 0, None) == 0; e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
gb = R * N * 4 / 1e9
print(f"minmax_fit {R}x{N}: {best*1e3:.0f} us incl. malloc+sync, {gb/best*1e3/1e3:.2f} TB/s ({gb:.2f} GB read)")
y = (src[:4].double() * sc.double() + of.double())
assert (dst.view(R, N)[:4].double() - y).abs().max() < 8e-3
