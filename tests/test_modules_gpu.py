"""The reference-shaped Python surface (modules.VAE_network.VAE, modules.train.train) on the GPU."""
import os
import random

import numpy as np
import pytest
import torch

import simulgen_vae_amd
from tests.gpu_common import G0, GOLD, make_cfg, relerr
from simulgen_vae_amd.init import synthetic_eps, synthetic_samples
from simulgen_vae_amd.spec import param_spec

simulgen_vae_amd.install_reference_api()
from modules.VAE_network import VAE  # noqa: E402
from modules import augmentation as aug  # noqa: E402
from modules import train as tr  # noqa: E402

pytestmark = pytest.mark.gpu


def test_vae_api_matches_reference_golden():
    """VAE.encoder / VAE.decoder(mode='fix') / state_dict round trip on the reference's step-3 state."""
    g = np.load(os.path.join(GOLD, "g0_small_MSE.npz"))
    cfg = make_cfg(G0)
    B = int(g["meta"][6])
    m = VAE(cfg.latent_dim, cfg.hierarchical_dim, cfg.num_filter_enc, cfg.num_filter_dec, cfg.num_node, cfg.num_time,
            lossfun="MSE", batch_size=B, small=True, compute_dtype="f32")
    sd = {e.name: torch.from_numpy(g["s3." + e.name]) for e in param_spec(cfg)}
    m.load_state_dict(sd)
    m.eval()
    x = torch.from_numpy(synthetic_samples(int(g["meta"][4]), range(100, 100 + B), cfg.num_node, cfg.num_time))
    mu, lv, xs = m.encoder(x)
    assert relerr(mu.cpu().numpy(), g["eval.mu"]) < 3e-4 and relerr(lv.cpu().numpy(), g["eval.log_var"]) < 3e-4
    assert len(xs) == 3 and relerr(xs[0].cpu().numpy(), g["eval.xs0"]) < 3e-4
    eps = synthetic_eps(int(g["meta"][5]), 100, cfg, B)
    m._eng().set_eps([torch.from_numpy(eps[0]).cuda()] + [torch.from_numpy(e).cuda() for e in eps[1:]])
    xf, kls = m.decoder(torch.from_numpy(g["fix.z"]), xs, mode="fix")
    assert relerr(xf.cpu().numpy(), g["fix.x_hat"]) < 3e-4 and len(kls) == 2
    out = m.state_dict()
    assert list(out.keys()) == [e.name for e in param_spec(cfg)]
    for k in ("decoder.recon.0.weight_orig", "decoder.decoder_blocks.1.module_list.0._seq.0.weight_orig",
              "encoder.xs_linear.2.weight_v", "decoder.sequence_start.0.0.bias"):
        np.testing.assert_array_equal(out[k].numpy(), g["s3." + k])
    x_hat, recon, kl_list, mse = m(x)
    assert x_hat.shape == (B, cfg.num_node, cfg.num_time) and len(kl_list) == 3
    assert abs(float(recon) - float(mse)) < 1e-9


def test_train_plumbing_run(tmp_path, monkeypatch):
    """BASELINE.json configs[0]-style plumbing through modules.train.train on synthetic data (8 params,
    4 epochs: the smallest the reference itself accepts), then reload the saved model."""
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    random.seed(0)
    np.random.seed(0)
    enc = [64, 32, 16, 8]
    N, T, P, B, E = 512, 16, 8, 4, 4
    x = synthetic_samples(20251003, range(P), N, T)
    tl, vl = aug.create_augmented_dataloaders(x, B, load_all=True)
    out = tr.train(E, B, tl, vl, 1e-3, enc, enc[::-1], N, 32, 8, T, 1e6, "MSE", True, True, compute_dtype="f32")
    loss, recon, kl, val = out
    assert all(len(a) == E for a in out) and np.isfinite(np.concatenate(out)).all()
    assert val[0] > 0 and val[-1] > 0 and val[1] == val[0]           # validation at epoch 0 and the last only
    assert loss[-1] < loss[0]
    assert os.path.exists("checkpoints/SimulGen-VAE.pth") and os.path.exists("model_save/SimulGen-VAE")
    sd = torch.load("checkpoints/SimulGen-VAE.pth", weights_only=True)
    assert len(sd) == 240
    m2 = torch.load("model_save/SimulGen-VAE", weights_only=False)    # our own file (pickled mirror object)
    m2.eval()
    m1 = tr.train.last_model
    m1.eval()
    xb = torch.from_numpy(x[:B])
    mu1 = m1.encoder(xb)[0].cpu().numpy()
    mu2 = m2.encoder(xb)[0].cpu().numpy()
    np.testing.assert_allclose(mu1, mu2, rtol=1e-4, atol=1e-5)   # split-K heads use float atomics: order varies
    with pytest.raises(ValueError):
        tr.train(2, B, tl, vl, 1e-3, enc, enc[::-1], N, 32, 8, T, 1e6, "MSE", True, True)   # epochs < 4
