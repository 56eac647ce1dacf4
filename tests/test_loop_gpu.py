"""Loop-level parity: the mirror of modules.train.train against the per-epoch arrays the REFERENCE's own train() produced
(tests/golden/gen_loop_fixtures.py -> loop_train.npz: BASELINE.json configs[0]'s shape with the minimum sizes the reference's
loop accepts -- preset filters, N = 4096, T = 32, batch 4, 8 + 4 samples, 4 epochs).

The run is replayed with the same Philox initial weights, the same batches and the same injected reparameterisation noise
(the f-th forward of the run, training or validation, gets synthetic_eps(EPS_SEED, f, ...)).  Eight AdamW steps at lr 1e-3
amplify rounding differences; stated tolerances (measured in parentheses): fp32 engine losses 5e-5 (3.5e-6), KL 1e-4 (1.3e-5),
average gradient norm 1e-3 (2.5e-4, from the 5-digit log line); bf16 engine losses 5e-3 (1.6e-3), KL 1e-2 (2.4e-3), gradient
norm 5e-3 (1.5e-3); beta and the learning rate are exact schedule values (1e-12)."""
import logging
import os
import re

import numpy as np
import pytest
import torch

import simulgen_vae_amd  # noqa: F401
from simulgen_vae_amd import engine as E
from simulgen_vae_amd.init import synthetic_eps, synthetic_samples
from simulgen_vae_amd.spec import VAEConfig

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ENC = [1024, 512, 256, 128]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_train_loop_matches_reference_run(dtype, tmp_path, monkeypatch):
    from simulgen_vae_amd.modules import train as T
    g = np.load(os.path.join(GOLD, "loop_train.npz"))
    N, Tn, B, EPOCHS, alpha, LR, eps_seed, data_seed, n_fwd = g["meta"]
    N, Tn, B, EPOCHS = int(N), int(Tn), int(B), int(EPOCHS)
    cfg = VAEConfig(32, 8, ENC, ENC[::-1], N, Tn, "MSE", True)
    train_batches = [torch.from_numpy(synthetic_samples(int(data_seed), range(i * B, (i + 1) * B), N, Tn)) for i in range(2)]
    val_batches = [torch.from_numpy(synthetic_samples(int(data_seed), range(100, 100 + B), N, Tn))]
    counter = {"f": 0}
    real_forward = E.Engine.forward

    def forward(self, train=True, fix=False, sync=True):
        eps = synthetic_eps(int(eps_seed), counter["f"], cfg, B)       # the f-th forward of the run, as the fixture generator
        counter["f"] += 1
        self.set_eps([torch.from_numpy(e).cuda() for e in eps])
        return real_forward(self, train=train, fix=fix, sync=sync)

    monkeypatch.setattr(E.Engine, "forward", forward)
    monkeypatch.chdir(tmp_path)
    lines = []

    class H(logging.Handler):
        def emit(self, rec):
            lines.append(rec.getMessage())

    h = H()
    logging.getLogger().addHandler(h)
    old_level = logging.getLogger().level
    logging.getLogger().setLevel(logging.INFO)
    try:
        loss, recon, kl, val = T.train(EPOCHS, B, train_batches, val_batches, float(LR), ENC, ENC[::-1], N, 32, 8, Tn, float(alpha),
                                       "MSE", True, True, compute_dtype=dtype)
    finally:
        logging.getLogger().removeHandler(h)
        logging.getLogger().setLevel(old_level)
    assert counter["f"] == int(n_fwd)                      # same number of forwards (training + validation) as the reference ran
    beta, lr, avg = [], [], []
    for ln in lines:
        m = re.search(r"Beta:([0-9.E+-]+)\s+AvgGrad:([0-9.E+-]+).*LR: ([0-9.E+-]+)", ln)
        if m:
            beta.append(float(m.group(1))); avg.append(float(m.group(2))); lr.append(float(m.group(3)))
    assert len(beta) == EPOCHS
    np.testing.assert_allclose(beta, g["beta"], rtol=1e-12)
    np.testing.assert_allclose(lr, g["lr"], rtol=1e-12)
    tl, tk, tg = (5e-5, 1e-4, 1e-3) if dtype == "f32" else (5e-3, 1e-2, 5e-3)
    print(f"[{dtype}] loss {loss} vs {g['loss']}\n kl {kl} vs {g['kl']}\n val {val} vs {g['val']}\n avg grad {avg} vs {g['avg_grad']}")
    np.testing.assert_allclose(loss, g["loss"], rtol=tl)
    np.testing.assert_allclose(recon, g["recon"], rtol=tl)
    np.testing.assert_allclose(val, g["val"], rtol=tl)
    np.testing.assert_allclose(kl, g["kl"], rtol=tk)
    np.testing.assert_allclose(avg, g["avg_grad"], rtol=tg)
    assert os.path.exists("checkpoints/SimulGen-VAE.pth") and os.path.exists("model_save/SimulGen-VAE")
    T.train.last_model._engine.close()
