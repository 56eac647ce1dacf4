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


def lc_init_state(shapes: dict, seed: int) -> dict:
    """Seeded state for the image latent conditioner, keyed like its state_dict (`shapes`: name -> shape): He-uniform
    weights, unit-norm spectral-norm vectors, norm scales near one, small biases, fresh BatchNorm buffers.  Used by the
    loop fixtures (tests/golden/gen_lc_loop_fixtures.py loads it into the REFERENCE model, tests/test_lc_loop_gpu.py into
    the mirror), so that both start from the same numbers without the fixture carrying any weights."""
    out = {}
    for idx, name in enumerate(sorted(shapes)):
        shape = tuple(int(d) for d in shapes[name])
        rng = _rng(seed, 5000 + idx)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, np.int64)
        elif name.endswith("running_mean"):
            out[name] = np.zeros(shape, np.float32)
        elif name.endswith("running_var"):
            out[name] = np.ones(shape, np.float32)
        elif name.endswith("weight_u") or name.endswith("weight_v"):
            v = rng.standard_normal(shape)
            out[name] = (v / np.linalg.norm(v)).astype(np.float32)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            b = float(np.sqrt(6.0 / fan_in))
            out[name] = rng.uniform(-b, b, shape).astype(np.float32)
        elif name.endswith("bias"):
            out[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        else:
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
    return out


def lc_synthetic(seed: int, P: int, pixels: int, latent_dim_end: int, size2: int, latent_dim: int):
    """Synthetic conditioner data set: images U[0,1) [P, pixels], main latents N(0, .5) [P, latent_dim_end], hierarchical
    latents N(0, .5) [P, size2, latent_dim]."""
    x = _rng(seed, 1).random((P, pixels), dtype=np.float32)
    y1 = (0.5 * _rng(seed, 2).standard_normal((P, latent_dim_end))).astype(np.float32)
    y2 = (0.5 * _rng(seed, 3).standard_normal((P, size2, latent_dim))).astype(np.float32)
    return x, y1, y2


def noise_call(seed: int, k: int, shape) -> np.ndarray:
    """The k-th standard-normal draw of a run with injected noise (flat order, any shape of that size)."""
    n = int(np.prod(shape))
    return _rng(seed, 100000 + k).standard_normal(n).astype(np.float32).reshape(tuple(shape))
