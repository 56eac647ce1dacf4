#!/usr/bin/env python3
"""Diagnostic for the two-rank data-parallel test: R repetitions of the 3-step run of tests/test_modules_gpu.py::_two_rank_worker
(gloo, both ranks on cuda:0) against the in-process expectation, compared BITWISE per state_dict entry: which side differs
(rank 0 vs rank 1 vs expectation), in which tensors, by how much.   python tests/micro/ddp_flake_diag.py [R] [payload] [ahead]"""
import os, socket, sys, tempfile
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import simulgen_vae_amd  # noqa
import torch.multiprocessing as mp
from tests.test_modules_gpu import _two_rank_worker
from tests.gpu_common import G1, make_cfg
from simulgen_vae_amd.engine import Engine
from simulgen_vae_amd.init import init_state, synthetic_samples
from modules.train import _DevArray


def expectation(payload):
    cfg = make_cfg(G1); B = 2
    engs, flats, xs = [], [], []
    for r in range(2):
        e = Engine(cfg, max_batch=B, compute_dtype="f32"); e.load_state(init_state(cfg, 11, reference_init=True)); e.seed(100 + r)
        ptr, n = e.grad_buffer(); engs.append(e); flats.append(torch.as_tensor(_DevArray(ptr, n), device="cuda"))
        xs.append(torch.from_numpy(synthetic_samples(5, range(r * B, (r + 1) * B), cfg.num_node, cfg.num_time)).cuda())
    ranges = []
    engs[0].set_bucket_callback(lambda b, off, cnt: ranges.append((b, off, cnt)))
    for step in range(3):
        for e, x in zip(engs, xs):
            e.set_input(x); e.forward(train=True); e.backward(1e6, 1e-4)
        if step == 0:
            engs[0].set_bucket_callback(None)
        mean = (flats[0] + flats[1]) * 0.5
        if payload == "bf16":
            small = engs[0].bucket_count() - 1
            for b, off, cnt in ranges:
                if b != small:
                    sl = slice(off, off + cnt); mean[sl] = ((flats[0][sl].bfloat16() + flats[1][sl].bfloat16()) * 0.5).float()
        for e, f in zip(engs, flats):
            f.copy_(mean); e.adamw_step(1e-3)
    torch.cuda.synchronize()
    want = engs[0].state_dict()
    for e in engs:
        e.close()
    return want


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    payload = sys.argv[2] if len(sys.argv) > 2 else "f32"
    ahead = len(sys.argv) > 3 and sys.argv[3] == "1"
    wants = [expectation(payload) for _ in range(2)]
    same = all(np.array_equal(wants[0][k], wants[1][k]) for k in wants[0])
    print(f"expectation replays bitwise: {same}", flush=True)
    for rep in range(R):
        with tempfile.TemporaryDirectory() as d:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            mp.spawn(_two_rank_worker, args=(2, port, d, payload, ahead), nprocs=2, join=True)
            got = [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(2)]
        d01, d0w = [], []
        for k, w in wants[0].items():
            a, b = got[0][k.replace(".", "__")], got[1][k.replace(".", "__")]
            if not np.array_equal(a, b):
                d01.append((k, float(np.abs(a.astype(np.float64) - b).max())))
            if not np.array_equal(a, w):
                d0w.append((k, float(np.abs(a.astype(np.float64) - w).max() / (np.abs(w).max() + 1e-30))))
        print(f"rep {rep}: rank0 != rank1 in {len(d01)} tensors; rank0 != expectation in {len(d0w)} of {len(wants[0])} tensors", flush=True)
        for k, v in d0w[:6]:
            print(f"    {k}: max rel diff {v:.3e}")


if __name__ == "__main__":
    main()
