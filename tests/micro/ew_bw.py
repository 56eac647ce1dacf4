"""Bandwidth of the flat element-wise operator (ops.add) and a torch add on the same tensors: python tests/micro/ew_bw.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simulgen_vae_amd
from simulgen_vae_amd import ops
for mb in (8, 33.5, 67, 268):
    n = int(mb * 1e6 / 2) // 8 * 8
    a = torch.randn(n, device="cuda").to(torch.bfloat16); b = torch.randn(n, device="cuda").to(torch.bfloat16)
    for name, fn in (("ops.add", lambda: ops.add(a, b)), ("torch.add", lambda: torch.add(a, b))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{mb:6.1f} MB/tensor  {name:10s} {us:7.1f} us  {3 * n * 2 / us / 1e6:5.2f} TB/s")
