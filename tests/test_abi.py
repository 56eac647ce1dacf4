"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/sgvae.h
declares (no compute calls), and the param spec mirrors the reference state_dict recorded in the fixtures."""
import os
import re

import numpy as np

import simulgen_vae_amd  # noqa: F401
from simulgen_vae_amd import engine as E
from simulgen_vae_amd.spec import VAEConfig, num_params, param_spec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "sgvae.h")).read()
    declared = set(re.findall(r"\b(sgv_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"sgv_bucket_cb"}
    assert declared == set(E.ABI_SYMBOLS), declared ^ set(E.ABI_SYMBOLS)
    lib = E.load_library()
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.sgv_last_error() is not None


def test_library_exports_every_operator_symbol():
    """include/sgvae_ops.h (latent-conditioner operators) vs simulgen_vae_amd.ops.OPS_SYMBOLS vs the library."""
    from simulgen_vae_amd import ops
    hdr = open(os.path.join(ROOT, "include", "sgvae_ops.h")).read()
    declared = set(re.findall(r"\b(sgv_(?:op|pset)_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(ops.OPS_SYMBOLS), declared ^ set(ops.OPS_SYMBOLS)
    lib = E.load_library()
    for s in declared:
        assert hasattr(lib, s), s


def test_param_counts_match_reference_full_size():
    enc = [1024, 512, 256, 128]
    small = VAEConfig(32, 8, enc, enc[::-1], 95008, 200, "MSE", True)
    large = VAEConfig(32, 8, enc, enc[::-1], 95008, 200, "MSE", False)
    assert num_params(small) == 438161408      # SURVEY 8(a), measured on the reference
    assert num_params(large) == 496095488
    dead = sum(int(np.prod(e.shape)) for e in param_spec(small)
               if not e.trainable and e.kind in ("bias", "weight_orig", "gn_weight", "gn_bias"))
    assert dead == 36517968
    assert len(param_spec(small)) == 240 and len(param_spec(large)) == 342


def test_no_gpu_means_loud_failure():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = VAEConfig(32, 8, [32, 16, 8, 8], [8, 8, 16, 32], 72, 10)
    with pytest.raises(E.SgvError):
        E.Engine(cfg, max_batch=2)
