"""Split-K sweep of the small / mid-size layer GEMMs through the C ABI.  Run under
`rocprofv3 --kernel-trace`; tools-free post-processing: python tests/micro/splitk_sweep.py parse <kernel_trace.csv>.
Each (shape, splitk) is launched REPS times in a fixed order, so the n-th main-kernel record belongs to the n-th entry."""
import os, sys, csv
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REPS = 4
SKS = [1, 2, 3, 4, 5, 6, 8]
NT = [(512, 256, 3), (1024, 512, 3), (512, 512, 3), (512, 1024, 1), (256, 512, 3), (256, 1280, 1), (2560, 512, 1),
      (5120, 1024, 1), (512, 1024, 3), (512, 2560, 1), (1024, 1024, 3), (1024, 5120, 1), (1280, 1280, 5)]
TN = [(1024, 1024, 3), (1024, 512, 3), (512, 512, 3), (512, 256, 3), (256, 512, 1), (1024, 5120, 1), (1280, 1280, 5),
      (512, 1024, 1), (1280, 256, 1)]
M, T = 3200, 200


def plan():
    out = []
    for n, k, t in NT:
        for sk in SKS:
            out.append(("nt", n, k, t, sk))
    for n1, n2, t in TN:
        for sk in SKS:
            out.append(("tn", n1, n2, t, sk))
    return out


def main():
    import torch
    import simulgen_vae_amd
    from simulgen_vae_amd import engine as E
    lib = E.load_library()
    torch.manual_seed(0)
    for kind, a, b, t, sk in plan():
        if kind == "nt":
            A = torch.randn(M, b, device="cuda").to(torch.bfloat16)
            W = (torch.randn(t, a, b, device="cuda") * 0.05).to(torch.bfloat16)
            C = torch.empty(M, a, device="cuda", dtype=torch.bfloat16)
            for _ in range(REPS):
                assert lib.sgv_test_gemm_nt(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), None, None, None, M, a, b, t, T, sk, 0, None) == 0
        else:
            dY = torch.randn(M, a, device="cuda").to(torch.bfloat16)
            X = torch.randn(M, b, device="cuda").to(torch.bfloat16)
            out = torch.empty(t, a, b, device="cuda", dtype=torch.float32)
            for _ in range(REPS):
                assert lib.sgv_test_gemm_tn(1, dY.data_ptr(), X.data_ptr(), out.data_ptr(), M, a, b, t, T, sk, 1, None) == 0
    torch.cuda.synchronize()


def parse(path):
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ker = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows
           if "gemm_" in r["Kernel_Name"] or r["Kernel_Name"].startswith("sum_slabs")]
    i = 0
    table = {}
    for kind, a, b, t, sk in plan():
        best_main, best_red, name = 1e9, 0.0, ""
        for _ in range(REPS):
            name, d = ker[i]; i += 1
            assert "reduce" not in name and "sum_slabs" not in name, (name, kind, a, b, t, sk)
            red = 0.0
            if i < len(ker) and ("reduce" in ker[i][0] or "sum_slabs" in ker[i][0]):
                red = ker[i][1]; i += 1
            if d + red < best_main + best_red:
                best_main, best_red = d, red
        table.setdefault((kind, a, b, t), []).append((sk, best_main, best_red, name.split("(")[0][:28]))
    for (kind, a, b, t), v in table.items():
        flops = 2.0 * M * a * b * t
        best = min(v, key=lambda e: e[1] + e[2])
        print(f"{kind} {a:5d} {b:5d} taps={t}  {v[0][3]:28s} best sk={best[0]} {best[1]+best[2]:6.1f} us ({flops/(best[1]+best[2])/1e6:5.0f} TF/s) | " +
              " ".join(f"{sk}:{m:.0f}+{r:.0f}" for sk, m, r, _ in v))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "parse":
        parse(sys.argv[2])
    else:
        main()
