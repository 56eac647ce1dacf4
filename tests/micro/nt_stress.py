import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import test_kernels_gpu as tk
from simulgen_vae_amd import engine as E
lib = E.load_library()
dtype = int(sys.argv[1]); reps = int(sys.argv[2])
shapes = [(520, 320, 320, 5, 40, 2), (520, 320, 320, 5, 40, 1), (640, 512, 1024, 1, 200, 1), (3200, 1024, 1024, 3, 200, 1), (384, 256, 2048, 1, 64, 3)]
tot_bad = 0
for case in shapes:
    M, N, K, taps, Tlen, splitk = case
    rng = np.random.default_rng(1)
    A = tk._bf16_round(rng.standard_normal((M, K)).astype(np.float32))
    W = tk._bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.1)
    bias = rng.standard_normal(N).astype(np.float32)
    dA, dW = tk._dev(A, dtype), tk._dev(W, dtype)
    dbias = torch.from_numpy(bias).cuda()
    ref = tk.ref_conv_nt(A, W, bias, 1.0, None, taps, Tlen)
    nbad = 0; worst = 0
    for it in range(reps):
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
        rc = lib.sgv_test_gemm_nt(dtype, dA.data_ptr(), dW.data_ptr(), out.data_ptr(), dbias.data_ptr(), None, None, M, N, K, taps, Tlen, splitk, 1, None)
        e = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
        worst = max(worst, e)
        if not (e < 1e-4): nbad += 1
    print(case, "dtype", dtype, "bad runs %d/%d" % (nbad, reps), "worst %.2e" % worst, flush=True)
    tot_bad += nbad
print("TOTAL BAD", tot_bad)
