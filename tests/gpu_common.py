"""Shared helpers for the GPU parity tests (engine vs oracle / golden fixtures)."""
import os

import numpy as np

import simulgen_vae_amd  # noqa: F401
from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples
from simulgen_vae_amd.spec import VAEConfig, param_spec

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G0 = dict(latent_dim=32, hierarchical_dim=8, enc=[32, 16, 8, 8], num_node=72, num_time=10)
G1 = dict(latent_dim=32, hierarchical_dim=8, enc=[64, 32, 16, 8], num_node=520, num_time=12)
G2 = dict(latent_dim=32, hierarchical_dim=8, enc=[256, 128, 64, 32], num_node=2080, num_time=40)


def make_cfg(cfgd, small=True, lossfun="MSE"):
    return VAEConfig(cfgd["latent_dim"], cfgd["hierarchical_dim"], cfgd["enc"], cfgd["enc"][::-1],
                     cfgd["num_node"], cfgd["num_time"], lossfun, small)


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def engine_step(eng, cfg, x, eps, alpha, beta, want_acts=True):
    """forward(train) + backward on the engine; returns scalars, activations, grads."""
    import torch
    B = x.shape[0]
    eng.set_input(torch.from_numpy(x).cuda())
    eng.set_eps([torch.from_numpy(e).cuda() for e in eps])
    sc = eng.forward(train=True)
    acts = {}
    if want_acts:
        T = cfg.num_time
        for i, c in enumerate(cfg.num_filter_enc):
            acts[f"enc_h{i}"] = eng.activation(f"enc_h{i}", (B, c, T))
        for i in range(len(cfg.num_filter_dec) - 1):
            acts[f"dec_out{i}"] = eng.activation(f"dec_out{i}", (B, cfg.num_filter_dec[i + 1], T))
        for i in range(len(cfg.num_filter_dec) - 2):
            acts[f"zmap{i}"] = eng.activation(f"zmap{i}", (B, cfg.num_filter_dec[i + 1], T))
        acts["mu"] = eng.activation("mu", (B, cfg.latent_dim))
        acts["log_var"] = eng.activation("log_var", (B, cfg.latent_dim))
        acts["x_hat"] = eng.activation("x_hat", (B, cfg.num_node, T))
    eng.backward(alpha, beta)
    return sc, acts


def oracle_step(orc, x, eps, alpha, beta):
    xhat, rl, kls, mse = orc.forward(x, eps)
    acts = dict(orc.acts)
    grads = orc.backward(alpha, beta)
    return dict(recon=float(rl), kls=[float(k) for k in kls], mse=float(mse)), acts, grads
