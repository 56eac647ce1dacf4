"""One-shot diagnostic for the GPU box: runs the kernel checks and an engine-vs-oracle step and
prints EVERY per-tensor error instead of stopping at the first failure.
    python tests/gpu_diag.py [g0|g1|g2] [f32|bf16] [small|large]
"""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_common import G0, G1, G2, engine_step, make_cfg, oracle_step, rel_l2, relerr  # noqa: E402
from simulgen_vae_amd import engine as E  # noqa: E402
from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402


def kernels():
    import torch
    from tests import test_kernels_gpu as tk
    for dtype in (0, 1):
        for case in tk.NT_CASES:
            try:
                tk.test_gemm_nt(dtype, case)
                print(f"gemm_nt dtype={dtype} {case}: ok")
            except BaseException as ex:  # noqa: BLE001
                print(f"gemm_nt dtype={dtype} {case}: FAIL {str(ex)[:200]}")
    for dtype, tr in ((0, 0), (1, 1), (1, 0)):
        for case in tk.TN_CASES:
            try:
                tk.test_gemm_tn(dtype, tr, case)
                print(f"gemm_tn dtype={dtype} tr={tr} {case}: ok")
            except BaseException as ex:  # noqa: BLE001
                print(f"gemm_tn dtype={dtype} tr={tr} {case}: FAIL {str(ex)[:200]}")


def engine(cfgname, dtype, small, B=3, steps=2):
    cfgd = dict(g0=G0, g1=G1, g2=G2)[cfgname]
    cfg = make_cfg(cfgd, small)
    state = init_state(cfg, 7)
    eng = E.Engine(cfg, max_batch=B, compute_dtype=dtype)
    eng.load_state(state)
    orc = vo.OracleVAE(cfg, state)
    alpha, beta, lr = 1e6, 1e-4, 1e-3
    for step in range(steps):
        x = synthetic_samples(20251003, range(step * B, (step + 1) * B), cfg.num_node, cfg.num_time)
        eps = synthetic_eps(1234, step, cfg, B)
        sc, acts = engine_step(eng, cfg, x, eps, alpha, beta)
        osc, oacts, ograds = oracle_step(orc, x, eps, alpha, beta)
        print(f"--- {cfgname} {dtype} small={small} step {step}")
        print("scalars engine:", sc)
        print("scalars oracle:", osc)
        for k, v in acts.items():
            print(f"  act {k:12s} maxrel {relerr(v, oacts[k]):.3e}  l2 {rel_l2(v, oacts[k]):.3e}")
        sd = eng.state_dict()
        worst = 0.0
        for name in sorted(k for k in sd if k.endswith("_u") or k.endswith("_v")):
            worst = max(worst, relerr(sd[name], orc.P[name]))
        print(f"  u/v after forward: worst maxrel {worst:.3e}")
        for name, g in ograds.items():
            eg = eng.grad(name)
            if g is None or eg is None:
                if not (g is None and eg is None):
                    print(f"  grad {name}: NONE MISMATCH engine={eg is None} oracle={g is None}")
                continue
            print(f"  grad {name:62s} maxrel {relerr(eg, g):.3e}  l2 {rel_l2(eg, g):.3e}")
        print(f"  grad_norm engine {eng.grad_norm():.6e} oracle {orc.grad_norm():.6e}")
        eng.adamw_step(lr)
        orc.adamw_step(lr)
        sd = eng.state_dict()
        worst, wname = 0.0, ""
        for name, v in sd.items():
            r = relerr(v, orc.P[name])
            if r > worst:
                worst, wname = r, name
        print(f"  params after AdamW: worst maxrel {worst:.3e} ({wname})")
    eng.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    if not args or args[0] == "kernels":
        kernels()
    else:
        try:
            engine(args[0], args[1] if len(args) > 1 else "f32", (args[2] if len(args) > 2 else "small") == "small")
        except BaseException:  # noqa: BLE001
            traceback.print_exc()
            sys.exit(1)
