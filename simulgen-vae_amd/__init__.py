"""simulgen-vae_amd: MI355X-native (gfx950) training engine for the SimulGen-VAE hot path.

Layout: csrc/ (HIP kernels + the C-ABI library libsgvae.so), engine.py (ctypes binding),
spec.py / init.py (state_dict mirror + deterministic init), modules/ (host-side mirror of the
reference's Python interface for this path).
"""
from .spec import VAEConfig, param_spec, layer_list  # noqa: F401

__all__ = ["VAEConfig", "param_spec", "layer_list"]
