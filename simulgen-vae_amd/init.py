"""Deterministic parameter / input generation from numpy Philox streams.

Used by bench.py (random-init weights of the named architecture: there is no network for
checkpoints), by the fixture generator and by the parity tests, so that no weight tensor has
to be shipped: the same (config, seed) gives the same bytes on any box.

Weights follow the reference's init *distribution* (modules/common.py:39-59: kaiming-uniform,
bound sqrt(6/fan_in)); biases / GroupNorm affine are drawn non-trivially (not the reference's
0 / 1) so that every gradient path is exercised by the parity tests.  u, v are normalised
normals as in torch's legacy spectral_norm (nn/utils/spectral_norm.py: SpectralNorm.apply).
"""
from __future__ import annotations

import numpy as np

from .spec import VAEConfig, param_spec


def _rng(seed: int, idx: int) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed, idx]))


def init_state(cfg: VAEConfig, seed: int = 7, reference_init: bool = False) -> dict:
    """name -> float32 ndarray for every state_dict key, in spec order."""
    state = {}
    for idx, e in enumerate(param_spec(cfg)):
        g = _rng(seed, idx)
        if e.kind == "weight_orig":
            fan_in = int(np.prod(e.shape[1:]))
            bound = np.sqrt(6.0 / fan_in)
            a = g.uniform(-bound, bound, e.shape)
        elif e.kind == "bias":
            a = np.zeros(e.shape) if reference_init else g.uniform(-0.05, 0.05, e.shape)
        elif e.kind == "gn_weight":
            a = np.ones(e.shape) if reference_init else 1.0 + g.uniform(-0.2, 0.2, e.shape)
        elif e.kind == "gn_bias":
            a = np.zeros(e.shape) if reference_init else g.uniform(-0.1, 0.1, e.shape)
        else:  # weight_u / weight_v
            a = g.standard_normal(e.shape)
            a = a / max(np.linalg.norm(a), 1e-12)
        state[e.name] = np.ascontiguousarray(a, dtype=np.float32)
    return state


def synthetic_samples(seed: int, indices, num_node: int, num_time: int) -> np.ndarray:
    """Synthetic dataset rows X[p] ~ U(-0.7, 0.7) in the reference's post-transpose layout
    [P, num_node, num_time] (SimulGen-VAE.py:282; MinMax target range data_preprocess.py:90).
    Each sample has its own Philox stream, so any subset can be produced on any rank."""
    out = np.empty((len(indices), num_node, num_time), dtype=np.float32)
    for j, p in enumerate(indices):
        out[j] = _rng(seed, int(p)).uniform(-0.7, 0.7, (num_node, num_time)).astype(np.float32)
    return out


def synthetic_eps(seed: int, step: int, cfg: VAEConfig, batch: int):
    """The three reparameterisation noise tensors in the reference's draw order
    ([B,latent], [B,C2,T], [B,C1,T]; layouts [B,C,T])."""
    dec = cfg.num_filter_dec
    shapes = [(batch, cfg.latent_dim)] + [(batch, dec[i + 1], cfg.num_time) for i in range(len(dec) - 2)]
    return [_rng(seed, step * 16 + i).standard_normal(s).astype(np.float32) for i, s in enumerate(shapes)]
