"""Implicit-GEMM convolution microbenchmark (operator ABI): for a conditioner layer shape, the 2-D implicit kernels next to
the GEMMs on a materialised im2col matrix and the 1-D nine-tap GEMM of the same size.
   python tests/micro/conv_bench.py B H W Cin Cout [stride] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simulgen_vae_amd
from simulgen_vae_amd import ops, engine as E

B, H, W, Ci, Co = (int(v) for v in sys.argv[1:6])
S = int(sys.argv[6]) if len(sys.argv) > 6 else 1
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
lib = E.load_library()
torch.manual_seed(0)
bf = torch.bfloat16
x = torch.randn(B, H, W, Ci, device="cuda").to(bf)
wp = (torch.randn(Co, 9 * Ci, device="cuda") * 0.05).to(bf)
Ho, Wo = (H + 2 - 3) // S + 1, (W + 2 - 3) // S + 1
dy = torch.randn(B, Ho, Wo, Co, device="cuda").to(bf)
wt = wp.t().contiguous()
M = B * Ho * Wo
col = torch.randn(M, 9 * Ci, device="cuda").to(bf)
w1d = wp.view(Co, 9, Ci).permute(1, 0, 2).contiguous()         # [taps][N][K]
c1d = torch.empty(M, Co, device="cuda", dtype=bf)
flops = 2.0 * M * Co * 9 * Ci


def t(name, fn, fl=flops):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    print(f"{name:34s} {best * 1e3:8.0f} us  {fl / best / 1e9:7.0f} TFLOP/s", flush=True)


print(f"B={B} {H}x{W} Cin={Ci} Cout={Co} stride={S}: M={M} K={9 * Ci}")
t("conv2d_nt forward (implicit)", lambda: ops.conv2d_nt(x, wp, Co, 3, 3, S, 1, 9 * Ci, Ci))
t("gemm_nt on an im2col matrix", lambda: ops.gemm_nt(col, wp))
if S == 1:
    def nt1d():
        rc = lib.sgv_test_gemm_nt256(x.data_ptr(), w1d.data_ptr(), c1d.data_ptr(), None, None, None, M, Co, Ci, 9, H * W, 1, 0, 1, 0, None, None, None)
        assert rc == 0, lib.sgv_last_error()
    if Co >= 256:
        t("1-D nine-tap GEMM, planner", nt1d)
    t("conv2d_nt input gradient (flip)", lambda: ops.conv2d_nt(dy, wt, Ci, 3, 3, 1, 1, Co, Ci * Co, flip=True))
t("conv2d_tn weight gradient", lambda: ops.conv2d_tn(dy, x, 3, 3, S, 1))
t("gemm_tn on an im2col matrix", lambda: ops.gemm_tn(dy.view(M, Co), col))
