import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import test_kernels_gpu as tk
from simulgen_vae_amd import engine as E
lib = E.load_library()
dtype = int(sys.argv[1])
base = (520, 320, 320, 5, 40, 1)
variants = [base, (640, 320, 320, 5, 40, 1), (520, 256, 320, 5, 40, 1), (520, 320, 256, 5, 40, 1), (520, 320, 512, 5, 40, 1), (520, 320, 320, 1, 40, 1),
            (520, 320, 320, 3, 40, 1), (520, 320, 320, 5, 65, 1), (520, 320, 320, 5, 520, 1), (120, 320, 320, 5, 40, 1), (128, 256, 320, 1, 128, 1),
            (128, 256, 1600, 1, 128, 1), (128, 256, 320, 5, 128, 1), (128, 256, 64, 5, 128, 1), (128, 256, 64, 25, 128, 1)]
for case in variants:
    M, N, K, taps, Tlen, splitk = case
    rng = np.random.default_rng(1)
    A = tk._bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = tk._bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.1)
    dA, dW = tk._dev(A, dtype), tk._dev(W, dtype)
    ref = tk.ref_conv_nt(A, W, None, 1.0, None, taps, Tlen)
    nbad = 0; rows = set()
    for it in range(8):
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        rc = lib.sgv_test_gemm_nt(dtype, dA.data_ptr(), dW.data_ptr(), out.data_ptr(), None, None, None, M, N, K, taps, Tlen, splitk, 1, None)
        e = np.abs(out.cpu().numpy() - ref) / np.abs(ref).max()
        if not (e.max() < 1e-4):
            nbad += 1; rows |= set((np.argwhere(e > 1e-4)[:, 0] % 128).tolist())
    print(case, "bad %d/8" % nbad, "local rows", sorted(rows)[:16], flush=True)
