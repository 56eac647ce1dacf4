"""hipBLASLt (csrc/vendor.hip) against the hand-written NT kernels on the one-tap layer shapes of the step, through the C ABI.
Run under `rocprofv3 --kernel-trace`, then `python tests/micro/vendor_vs_own.py parse <kernel_trace.csv>`: per shape the
kernel time of the own path (main kernel + split-K combine, split chosen as the engine does) and of the library path."""
import os, sys, csv
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REPS = 5
M = 3200
# N, K, split-K the engine picks for its own kernel
SHAPES = [(95008, 1024, 1), (5120, 1024, 1), (1024, 5120, 2), (2560, 512, 1), (512, 2560, 3), (1280, 256, 1), (256, 1280, 3),
          (1024, 512, 1), (512, 1024, 2), (1024, 1024, 1), (2048, 2048, 1), (512, 512, 1), (512, 256, 1), (256, 512, 1),
          (256, 128, 1), (128, 256, 1), (1024, 95008, 5)]


def main():
    import torch
    import simulgen_vae_amd
    from simulgen_vae_amd import engine as E
    lib = E.load_library()
    torch.manual_seed(0)
    for N, K, sk in SHAPES:
        A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        scale = torch.tensor([0.5], device="cuda")
        for _ in range(REPS):
            assert lib.sgv_test_gemm_nt(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), scale.data_ptr(), None, M, N, K, 1, 200, sk, 0, None) == 0
        for _ in range(REPS):
            rc = lib.sgv_test_gemm_nt_lib(A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), scale.data_ptr(), None, M, N, K, None)
            if rc != 0:
                print("lib refused", N, K, lib.sgv_last_error())
                break
    torch.cuda.synchronize()


def parse(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    ker = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
    ker = [k for k in ker if "gemm_nt" in k[0] or k[0].startswith("Cijk") or "Cijk" in k[0]]
    i = 0
    for N, K, sk in SHAPES:
        own = []
        for _ in range(REPS):
            d = ker[i][1]; i += 1
            if i < len(ker) and "reduce" in ker[i][0]:
                d += ker[i][1]; i += 1
            own.append(d)
        libt, name = [], ""
        while i < len(ker) and "Cijk" in ker[i][0] and len(libt) < REPS:
            libt.append(ker[i][1]); name = ker[i][0]; i += 1
        fl = 2.0 * M * N * K
        o, l = min(own), (min(libt) if libt else float("nan"))
        print(f"N={N:6d} K={K:5d}  own(sk={sk}) {o:7.1f} us {fl/o/1e6:6.0f} TF/s   lib {l:7.1f} us {fl/l/1e6:6.0f} TF/s   {name[:70]}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "parse":
        parse(sys.argv[2])
    else:
        main()
