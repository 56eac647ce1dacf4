"""Loop-level fixture: the REFERENCE's own modules.train.train (modules/train.py:50-256) run end to end on CPU.

Run only in the build container (needs /root/reference):

    python tests/golden/gen_loop_fixtures.py          # about a minute, writes tests/golden/loop_train.npz

Shape: BASELINE.json configs[0] with the minimum the reference's loop accepts: preset filters
[1024, 512, 256, 128], N = 4096, T = 32, batch 4, 8 training + 4 validation samples, 4 epochs (so the cosine-warm-restart
scheduler has T_0 = 1 and the beta warm-up covers epochs [1, 3)).  Everything random is pinned without touching the
reference's arithmetic:
  * initial weights: the reference draws them with torch's global RNG inside train(); `add_sn` is wrapped so that, after the
    real add_sn has run on the root module, the Philox state of simulgen_vae_amd.init.init_state(cfg, 7, reference_init=True)
    is loaded (the mirror's VAE() starts from the same state);
  * reparameterisation noise: torch.randn_like is replaced by an injector that serves synthetic_eps(EPS_SEED, f, cfg, B)
    for the f-th forward of the run (training and validation forwards in loop order);
  * data: plain lists of [4, N, T] tensors stand in for the loaders (no shuffling, no augmentation).
Recorded: the four returned per-epoch arrays, and beta / learning rate / average gradient norm per epoch parsed from the
reference's own log lines.
"""
import logging
import os
import re
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fixtures as gf  # noqa: E402  (stand-in modules + reference import path)
from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples  # noqa: E402
from simulgen_vae_amd.spec import VAEConfig  # noqa: E402

import modules.train as ref_train  # noqa: E402
from modules.VAE_network import VAE as RefVAE  # noqa: E402

ENC = [1024, 512, 256, 128]
N, T, B, EPOCHS = 4096, 32, 4, 4
ALPHA, LR = 1e6, 1e-3
EPS_SEED, DATA_SEED = 4321, 20251003


def main():
    cfg = VAEConfig(32, 8, ENC, ENC[::-1], N, T, "MSE", True)
    state = init_state(cfg, 7, reference_init=True)
    real_add_sn = ref_train.add_sn

    def add_sn_then_load(m):
        r = real_add_sn(m)
        if isinstance(m, RefVAE):
            m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
        return r

    class Inj:
        f = 0
        queue = []

        def __call__(self, t):
            if not self.queue:
                self.queue = [torch.from_numpy(e) for e in synthetic_eps(EPS_SEED, self.f, cfg, B)]
                Inj.f = self.f = self.f + 1
            e = self.queue.pop(0)
            assert tuple(e.shape) == tuple(t.shape), (e.shape, t.shape)
            return e.to(t.dtype)

    inj = Inj()
    train_batches = [torch.from_numpy(synthetic_samples(DATA_SEED, range(i * B, (i + 1) * B), N, T)) for i in range(2)]
    val_batches = [torch.from_numpy(synthetic_samples(DATA_SEED, range(100, 100 + B), N, T))]
    lines = []

    class H(logging.Handler):
        def emit(self, rec):
            lines.append(rec.getMessage())

    h = H()
    logging.getLogger().addHandler(h)
    logging.getLogger().setLevel(logging.INFO)
    ref_train.add_sn = add_sn_then_load
    torch.randn_like = inj
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)
            os.makedirs("model_save", exist_ok=True)
            loss, recon, kl, val = ref_train.train(EPOCHS, B, train_batches, val_batches, LR, ENC, ENC[::-1], N, 32, 8, T, ALPHA,
                                                   "MSE", True, True)
            os.chdir(cwd)
    finally:
        os.chdir(cwd)
        ref_train.add_sn = real_add_sn
        torch.randn_like = gf._REAL_RANDN_LIKE
        logging.getLogger().removeHandler(h)
    beta, lr, avg = [], [], []
    for ln in lines:
        m = re.search(r"Beta:([0-9.E+-]+)\s+AvgGrad:([0-9.E+-]+).*LR: ([0-9.E+-]+)", ln)
        if m:
            beta.append(float(m.group(1))); avg.append(float(m.group(2))); lr.append(float(m.group(3)))
    assert len(beta) == EPOCHS, lines
    out = dict(loss=loss, recon=recon, kl=kl, val=val, beta=np.array(beta), lr=np.array(lr), avg_grad=np.array(avg),
               meta=np.array([N, T, B, EPOCHS, ALPHA, LR, EPS_SEED, DATA_SEED, inj.f], dtype=np.float64))
    path = os.path.join(HERE, "loop_train.npz")
    np.savez_compressed(path, **out)
    print("loop_train:", {k: np.asarray(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
