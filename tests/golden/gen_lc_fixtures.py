#!/usr/bin/env python3
"""Golden vectors for the image latent conditioner, recorded from the REFERENCE model
(modules/latent_conditioner_model_cnn.py::LatentConditionerImg, imported from /root/reference) on a small config:
the reference's own initialisation (state_dict incl. spectral-norm u/v), one eval forward, one training forward with
the dropout masks captured, the loss of the training loop (latent_conditioner.py:293-296: 10*MSE(y1) + MSE(y2)), every
gradient, the BatchNorm running buffers and spectral-norm vectors after the step, the clipped gradient norm and the
parameters after one AdamW step (latent_conditioner.py:196,304,314).  Run here only; the fixture travels, the reference does not.
    python tests/golden/gen_lc_fixtures.py"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from modules.latent_conditioner_model_cnn import LatentConditionerImg  # noqa: E402  (reference)

FILTERS = [16, 32, 32, 64, 64, 128]
LATENT_END, LATENT, SIZE2, IMG, B = 32, 8, 3, 16, 4


def margins(m, x):
    """Smallest |ReLU input| and smallest top-2 gap of any max-pool window in a training forward: gradients of a
    piecewise-linear net are discontinuous where these vanish, and two correct fp32 implementations round differently
    there (a single flipped ReLU gate showed up as a 15 % max-norm gradient difference).  The fixture must stay clear."""
    lo = {"relu": float("inf"), "pool": float("inf")}
    o_relu, o_pool, o_drop = F.relu, F.max_pool2d, F.dropout

    def relu(inp, inplace=False):
        lo["relu"] = min(lo["relu"], float(inp.detach().abs().min()))
        return o_relu(inp, inplace)

    def pool(inp, *a, **k):
        xp = F.pad(inp.detach(), (1, 1, 1, 1), value=float("-inf"))
        win = xp.unfold(2, 3, 2).unfold(3, 3, 2).reshape(*xp.shape[:2], -1, 9)
        top = win.topk(2, dim=-1).values
        gap = top[..., 0] - top[..., 1]
        gap = gap[gap > 0]          # exact ties (windows of post-ReLU zeros) resolve to the first index in both implementations
        lo["pool"] = min(lo["pool"], float(gap.min()))
        return o_pool(inp, *a, **k)
    F.relu, F.max_pool2d = relu, pool
    F.dropout = lambda inp, p=0.5, training=True, inplace=False: inp
    try:
        with torch.no_grad():
            m(x)
    finally:
        F.relu, F.max_pool2d, F.dropout = o_relu, o_pool, o_drop
    return lo


def main():
    torch.manual_seed(1234)
    m = LatentConditionerImg(FILTERS, LATENT_END, (1, IMG, IMG), LATENT, SIZE2, (IMG, IMG), dropout_rate=0.3, use_attention=True)
    out = {"meta": np.array([LATENT_END, LATENT, SIZE2, IMG, B]), "filters": np.array(FILTERS)}
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for k, v in sd0.items():
        out["s0." + k] = v.numpy()
    # pick the first data seed whose training forward keeps every ReLU input / max-pool decision away from a tie
    for data_seed in range(7, 600):
        g = torch.Generator().manual_seed(data_seed)
        x = torch.rand(B, IMG * IMG, generator=g)
        m.load_state_dict(sd0)
        m.train()
        lo = margins(m, x)
        if lo["relu"] > 5e-5 and lo["pool"] > 5e-5:
            break
    else:
        raise SystemExit("no seed with safe margins")
    print("data seed", data_seed, "margins", lo)
    out["data_seed"] = np.array([data_seed])
    out["margins"] = np.array([lo["relu"], lo["pool"]])
    y1 = torch.randn(B, LATENT_END, generator=g) * 0.5
    y2 = torch.randn(B, SIZE2, LATENT, generator=g) * 0.5
    out.update(x=x.numpy(), y1=y1.numpy(), y2=y2.numpy())
    m.load_state_dict(sd0)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x)
    out.update(eval_main=e1.numpy(), eval_xs=e2.numpy())
    # training forward with captured dropout masks
    m.load_state_dict(sd0)
    m.train()
    masks = []
    orig = F.dropout

    def rec_dropout(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        mask = (torch.rand(inp.shape, generator=g) >= p).float()
        masks.append(mask)
        return inp * mask / (1.0 - p)
    F.dropout = rec_dropout
    try:
        p1, p2 = m(x)
    finally:
        F.dropout = orig
    A = nn.MSELoss()(p1, y1)
    Bl = nn.MSELoss()(p2, y2)
    loss = A * 10 + Bl
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}
    total_norm = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=10.0)
    opt.step()
    out.update(train_main=p1.detach().numpy(), train_xs=p2.detach().numpy(), loss=np.array([loss.item(), A.item(), Bl.item()]),
               total_norm=np.array([float(total_norm)]))
    for i, mk in enumerate(masks):
        out[f"mask{i}"] = mk.numpy()
    for n, gr in grads.items():
        if gr is not None:
            out["g." + n] = gr.numpy()
    for k, v in m.state_dict().items():
        out["s1." + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "lc_small.npz"), **out)
    print("lc_small.npz:", len(sd0), "state keys,", len(masks), "dropout masks, loss", loss.item(), "norm", float(total_norm))


if __name__ == "__main__":
    main()
