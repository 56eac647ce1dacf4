"""simulgen-vae_amd: MI355X-native (gfx950) training engine for the SimulGen-VAE hot path.

Layout: csrc/ (HIP kernels + the C-ABI library libsgvae.so), engine.py (ctypes binding),
spec.py / init.py (state_dict mirror + deterministic init), modules/ (host-side mirror of the
reference's Python interface for this path).
"""
from .spec import VAEConfig, param_spec, layer_list  # noqa: F401

__all__ = ["VAEConfig", "param_spec", "layer_list"]


def install_reference_api():
    """Make `import modules.<name>` resolve to this package's mirror of the reference interface
    (modules.VAE_network.VAE, modules.train.train, modules.utils.parse_condition_file, ...), so scripts
    written against the reference (SimulGen-VAE.py) run against the MI355X engine unchanged."""
    import importlib
    import sys
    pkg = importlib.import_module(__name__ + ".modules")
    sys.modules["modules"] = pkg
    for sub in ("VAE_network", "train", "utils", "augmentation", "losses", "data_preprocess", "latent_conditioner_model_cnn",
                "latent_conditioner", "latent_conditioner_e2e", "reconstruction_evaluator"):
        sys.modules["modules." + sub] = importlib.import_module(__name__ + ".modules." + sub)
    return pkg
