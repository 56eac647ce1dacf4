import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import test_kernels_gpu as tk
from simulgen_vae_amd import engine as E
lib = E.load_library()
case = tuple(int(v) for v in sys.argv[1:7]); dtype = int(sys.argv[7])
M, N, K, taps, Tlen, splitk = case
rng = np.random.default_rng(1)
A = tk._bf16_round(rng.standard_normal((M, K)).astype(np.float32))
W = tk._bf16_round(rng.standard_normal((taps, N, K)).astype(np.float32) * 0.1)
bias = rng.standard_normal(N).astype(np.float32)
dA, dW = tk._dev(A, dtype), tk._dev(W, dtype)
dbias = torch.from_numpy(bias).cuda()
ref = tk.ref_conv_nt(A, W, bias, 1.0, None, taps, Tlen)
for it in range(6):
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    rc = lib.sgv_test_gemm_nt(dtype, dA.data_ptr(), dW.data_ptr(), out.data_ptr(), dbias.data_ptr(), None, None, M, N, K, taps, Tlen, splitk, 1, None)
    got = out.cpu().numpy()
    e = np.abs(got - ref) / np.abs(ref).max()
    bad = np.argwhere(e > 1e-4)
    print(it, "rc", rc, "maxerr %.3e" % e.max(), "nbad", len(bad), "first bad", bad[:4].tolist(), "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:12])
