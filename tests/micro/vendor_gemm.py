"""What the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the plain-GEMM layers of the step, for
comparison with the hand-written kernels (printed beside each shape: the in-step figure from profiles/).  bf16, fp32 accumulate."""
import torch
torch.manual_seed(0)
dev = "cuda"


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def nt(M, N, K, note):
    A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.05
    C = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    us = bench(lambda: torch.matmul(A, W.t(), out=C))
    print(f"NT  C[{M}x{N}] = A[{M}x{K}] W[{N}x{K}]^T : {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s   ({note})")


def tn(M, N1, N2, note):
    dY = torch.randn(M, N1, device=dev, dtype=torch.bfloat16)
    X = torch.randn(M, N2, device=dev, dtype=torch.bfloat16)
    out = torch.empty(N1, N2, device=dev, dtype=torch.float32)
    us = bench(lambda: torch.matmul(dY.t(), X, out=None))
    print(f"TN  dW[{N1}x{N2}] = dY[{M}x{N1}]^T X[{M}x{N2}] : {us:8.1f} us  {2.0*M*N1*N2/us/1e6:7.1f} TFLOP/s   ({note})")


nt(3200, 95008, 1024, "recon forward: gemm_nt_kernel 856 us / 727 TF/s")
nt(3200, 1024, 95008, "encoder layer 0 forward, recon dX: gemm_nt_wide64p 806-825 us / 755-772 TF/s")
nt(3200, 5120, 1024, "gemm_nt 55-57 us")
nt(3200, 1024, 5120, "gemm_nt_wide 57-63 us")
nt(3200, 5120, 25600, "same FLOPs as the 5120^2 k5 layer as a plain GEMM: gemm_nt_wide64p 861 us / 974 TF/s")
tn(3200, 95008, 1024, "recon dW: gemm_tn_w2 628 us / 990 TF/s (fp32 out)")
tn(3200, 1024, 95008, "encoder layer 0 dW: gemm_tn_w2 612 us / 1017 TF/s (fp32 out)")
