"""Standalone GEMM microbenchmark through the C ABI (for rocprofv3 counter passes).
   python tests/micro/gemm_bench.py {nt,nt256} M N K taps [reps]   |   tn M N1 N2 taps [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simulgen_vae_amd
from simulgen_vae_amd import engine as E
lib = E.load_library()
kind = sys.argv[1]
a, b, c, taps = (int(v) for v in sys.argv[2:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
T = 200
torch.manual_seed(0)
if kind in ("nt", "nt256"):
    M, N, K = a, b, c
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(taps, N, K, device="cuda") * 0.05).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    sk = int(os.environ.get("SPLITK", "1"))
    stats = int(os.environ.get("STATS", "0"))      # 1: GroupNorm(8) statistics from the epilogue (the recon head's launch: all row tiles, split-K 1)
    sums = torch.zeros(M // T * 8 * 2, device="cuda", dtype=torch.float64) if stats else None
    def run():
        if kind == "nt256":     # the planner's choice for the shape (256x256 persistent kernel + 128-row tail); the call allocates its workspaces
            rc = lib.sgv_test_gemm_nt256(A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, None, M, N, K, taps, T, sk, 0, 1,
                                         N // 8 if stats else 0, sums.data_ptr() if stats else None, None, None)
        else:
            rc = lib.sgv_test_gemm_nt(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, None, M, N, K, taps, T, sk, 0, None)
        assert rc == 0, lib.sgv_last_error()
    flops = 2.0 * M * N * K * taps
else:
    M, N1, N2 = a, b, c
    dY = torch.randn(M, N1, device="cuda").to(torch.bfloat16)
    X = torch.randn(M, N2, device="cuda").to(torch.bfloat16)
    out = torch.empty(taps, N1, N2, device="cuda", dtype=torch.float32)
    use_tr, sk = int(os.environ.get("USE_TR", "1")), int(os.environ.get("SPLITK", "1"))   # 2 = force the w2 kernel, 4 = the 256 x 256 kernel, 6 = with bf16 output, 7 = its work-stealing form
    def run():
        rc = lib.sgv_test_gemm_tn(1, dY.data_ptr(), X.data_ptr(), out.data_ptr(), M, N1, N2, taps, T, sk, use_tr, None)
        assert rc == 0, lib.sgv_last_error()
    flops = 2.0 * M * N1 * N2 * taps
run(); torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
# OCCUPY="blocks,threads,lds_bytes": that many workgroups spin on another stream for 4 ms while the GEMM runs (a collective's channels)
occ = [int(v) for v in os.environ["OCCUPY"].split(",")] if os.environ.get("OCCUPY") else None
side = torch.cuda.Stream() if occ else None
for _ in range(reps):
    if occ:
        assert lib.sgv_test_occupy(side.cuda_stream, occ[0], occ[1], occ[2], 400000) == 0, lib.sgv_last_error()
        time.sleep(0.0005)           # the occupying workgroups are resident before the GEMM is launched
    ev0.record(); run(); ev1.record(); torch.cuda.synchronize(); ts.append(ev0.elapsed_time(ev1))
best = min(ts)
print(f"{kind} {a}x{b}x{c} taps={taps}: best {best*1e3:.0f} us  {flops/best/1e9:.0f} TFLOP/s (incl. launch+sync overhead)")
