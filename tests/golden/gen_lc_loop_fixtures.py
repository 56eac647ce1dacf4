"""Loop-level fixtures for the two latent-conditioner loops: the REFERENCE's own
  modules.latent_conditioner.train_latent_conditioner         (latent_conditioner.py:213-385)   -> loop_lc.npz
  modules.latent_conditioner_e2e.train_latent_conditioner_e2e (latent_conditioner_e2e.py:213-561) -> loop_e2e.npz
run end to end on CPU (fp32) on tiny seeded data.

Run only in the build container (needs /root/reference):

    python tests/golden/gen_lc_loop_fixtures.py          # about half a minute

The reference is imported unmodified.  Besides the stand-ins of gen_fixtures.py, two image-reader imports that are not
installed here (cv2, natsort: used by the PNG readers only) and the TensorBoard writer (add_scalar must exist) are registered
as empty stand-ins.  Everything random is pinned without touching the reference's arithmetic:
  * weights: after the loop's own `latent_conditioner.apply(...)` re-initialisation, simulgen_vae_amd.init.lc_init_state
    (numpy Philox, keyed by state_dict name) is loaded into the model, so the fixture carries no weights;
  * data: simulgen_vae_amd.init.lc_synthetic; plain lists of batches stand in for the loaders;
  * augmentation of the plain loop off: is_image_data=False and torch.rand(1) (the mixup / noise coin flips) returns 0.99;
  * the end-to-end loop's Gaussian noise (input, target, both latent targets, every batch) and the frozen decoder's
    reparameterisation noise: torch.randn_like serves simulgen_vae_amd.init.noise_call(NOISE_SEED, k, shape) for the k-th call
    of the run;
  * Dropout (two p = 0.2 layers in the heads): keeps everything, scaled by 1/(1-p) (a mask of ones);
  * the frozen VAE of the end-to-end loop: the reference VAE (G1 sizes) with simulgen_vae_amd.init.init_state weights;
    `load_vae_model` / `load_scaler` are replaced by functions returning the prepared objects (no pickle is read).
Recorded: every value the loops' loss modules return, in call order (nn.MSELoss / nn.HuberLoss subclasses that record
their result), the gradient norm of every step, the per-epoch numbers parsed from the loops' own log lines, the loops' return
values, and norm + 64 sampled entries of every tensor of the final state_dict.
"""
import contextlib
import io
import os
import re
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
for name, attrs in [("cv2", {"INTER_CUBIC": 2}), ("natsort", {}),
                    ("torch.utils.tensorboard", {"SummaryWriter": type("SummaryWriter", (), {
                        "__init__": lambda s, *a, **k: None, "add_scalar": lambda s, *a, **k: None})})]:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
sys.path.insert(0, HERE)
import gen_fixtures as gf  # noqa: E402  (remaining stand-ins + reference import path)
from simulgen_vae_amd.init import lc_init_state, lc_synthetic, noise_call, synthetic_samples  # noqa: E402

import modules.latent_conditioner as ref_lc  # noqa: E402
import modules.latent_conditioner_e2e as ref_e2e  # noqa: E402
from modules.latent_conditioner_model_cnn import LatentConditionerImg as RefLC  # noqa: E402

FILTERS = [16, 32, 32, 64, 64, 128]
LATENT_END, LATENT, SIZE2, IMG, B = 32, 8, 3, 16, 4
P_TRAIN, P_VAL, EPOCHS = 8, 4, 3
STATE_SEED, DATA_SEED, NOISE_SEED = 31, 77, 909


def sample_positions(name, numel, n=64):
    seed = int.from_bytes(name.encode()[-8:].rjust(8, b"\0"), "little") % (2 ** 31)
    rng = np.random.Generator(np.random.Philox(key=[977, seed]))
    return rng.integers(0, numel, size=min(n, numel))


class Recorder:
    """Replaces a torch.nn loss class by a subclass that records every returned value."""

    def __init__(self, *classes):
        self.values, self.classes, self.saved = [], classes, {}

    def __enter__(self):
        rec = self.values
        for name in self.classes:
            base = getattr(nn, name)
            self.saved[name] = base

            def forward(self, a, b, _base=base, _name=name):
                out = _base.forward(self, a, b)
                rec.append((_name, float(out)))
                return out
            setattr(nn, name, type(name, (base,), {"forward": forward}))
        return self

    def __exit__(self, *a):
        for name, base in self.saved.items():
            setattr(nn, name, base)


def keep_all_dropout(inp, p=0.5, training=True, inplace=False):
    return inp * (1.0 / (1.0 - p)) if training and p > 0.0 else inp


def final_state(model, out):
    for k, v in model.state_dict().items():
        a = v.detach().double().numpy().reshape(-1)
        out["fnorm." + k] = np.array(np.linalg.norm(a))
        out["fsamp." + k] = a[sample_positions(k, a.size)]


def make_model():
    torch.manual_seed(1)
    m = RefLC(FILTERS, LATENT_END, (1, IMG, IMG), LATENT, SIZE2, (IMG, IMG), dropout_rate=0.0, use_attention=True)
    state = lc_init_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, STATE_SEED)
    return m, {k: torch.from_numpy(v.copy()) for k, v in state.items()}


def batches(arrs, lo, hi):
    return [tuple(torch.from_numpy(a[i:i + B]) for a in arrs) for i in range(lo, hi, B)]


def parse(lines, pattern):
    rows = []
    for ln in lines:
        m = re.search(pattern, ln)
        if m:
            rows.append([float(v) for v in m.groups()])
    return np.array(rows)


def run_in_tmp(fn):
    cwd = os.getcwd()
    buf = io.StringIO()
    try:
        with tempfile.TemporaryDirectory() as d:
            os.chdir(d)
            os.makedirs("model_save", exist_ok=True)
            os.makedirs("checkpoints", exist_ok=True)
            with contextlib.redirect_stdout(buf):
                ret = fn()
            os.chdir(cwd)
    finally:
        os.chdir(cwd)
    return ret, buf.getvalue().splitlines()


def plain_loop():
    m, state = make_model()
    x, y1, y2 = lc_synthetic(DATA_SEED, P_TRAIN + P_VAL, IMG * IMG, LATENT_END, SIZE2, LATENT)
    train, val = batches((x, y1, y2), 0, P_TRAIN), batches((x, y1, y2), P_TRAIN, P_TRAIN + P_VAL)
    real_he, real_rand, real_clip, real_drop = ref_lc.safe_initialize_weights_He, torch.rand, torch.nn.utils.clip_grad_norm_, F.dropout
    norms = []

    def he_then_load(mod):
        real_he(mod)
        if isinstance(mod, RefLC):
            mod.load_state_dict(state)

    def clip(params, max_norm, *a, **k):
        n = real_clip(params, max_norm, *a, **k)
        norms.append(float(n))
        return n

    ref_lc.safe_initialize_weights_He = he_then_load
    torch.rand = lambda *a, **k: torch.tensor([0.99]) if a == (1,) else real_rand(*a, **k)
    torch.nn.utils.clip_grad_norm_ = clip
    F.dropout = keep_all_dropout
    try:
        with Recorder("MSELoss") as rec:
            _, lines = run_in_tmp(lambda: ref_lc.train_latent_conditioner(EPOCHS, train, val, m, 1e-3, weight_decay=1e-4, is_image_data=False))
    finally:
        ref_lc.safe_initialize_weights_He, torch.rand, torch.nn.utils.clip_grad_norm_, F.dropout = real_he, real_rand, real_clip, real_drop
    epochs = parse(lines, r"Train: ([0-9.E+-]+) \(y1:([0-9.E+-]+), y2:([0-9.E+-]+)\), Val: ([0-9.E+-]+) \(y1:([0-9.E+-]+), y2:([0-9.E+-]+)\), LR: ([0-9.E+-]+)")
    assert len(epochs) == EPOCHS, lines
    out = dict(meta=np.array([LATENT_END, LATENT, SIZE2, IMG, B, P_TRAIN, P_VAL, EPOCHS, STATE_SEED, DATA_SEED], dtype=np.int64),
               filters=np.array(FILTERS), lr0=np.array(1e-3), wd=np.array(1e-4), mse=np.array([v for _, v in rec.values]),
               grad_norms=np.array(norms), epochs=epochs)
    final_state(m, out)
    np.savez_compressed(os.path.join(HERE, "loop_lc.npz"), **out)
    print("loop_lc: mse calls", len(rec.values), "norms", norms, "\n", epochs)


def e2e_loop():
    m, state = make_model()
    cfg, vae = gf.build(gf.CONFIGS["g1"], True, "MSE")        # reference VAE with init_state(cfg, gf.STATE_SEED) weights
    for p in vae.parameters():
        p.requires_grad = False
    vae.eval()
    P = P_TRAIN + P_VAL
    x, _, _ = lc_synthetic(DATA_SEED, P, IMG * IMG, LATENT_END, SIZE2, LATENT)
    rng = np.random.Generator(np.random.Philox(key=[DATA_SEED, 9]))
    lat = rng.standard_normal((P, cfg.latent_dim)) * 2.0
    xs = rng.standard_normal((P, SIZE2, cfg.hierarchical_dim)) * 0.5
    from sklearn.preprocessing import MinMaxScaler
    sc1, sc2 = MinMaxScaler(feature_range=(-0.7, 0.7)).fit(lat), MinMaxScaler(feature_range=(-0.7, 0.7)).fit(xs.reshape(P, -1))
    y1 = sc1.transform(lat).astype(np.float32)
    y2 = sc2.transform(xs.reshape(P, -1)).reshape(P, SIZE2, -1).astype(np.float32)
    target = synthetic_samples(5, range(P), cfg.num_node, cfg.num_time)
    train, val = batches((x, y1, y2, target), 0, P_TRAIN), batches((x, y1, y2, target), P_TRAIN, P)
    calls = {"k": 0}

    def randn_like(t):
        e = torch.from_numpy(noise_call(NOISE_SEED, calls["k"], tuple(t.shape)))
        calls["k"] += 1
        return e.to(t.dtype)

    real = (ref_e2e.load_vae_model, ref_e2e.load_scaler, torch.randn_like, F.dropout)
    orig_apply = RefLC.apply

    def apply_then_load(self, fn):          # class-level (the loop pickles the model object at its end)
        orig_apply(self, fn)
        if isinstance(self, RefLC):
            self.load_state_dict(state)
        return self
    RefLC.apply = apply_then_load
    ref_e2e.load_vae_model = lambda path, device: vae
    ref_e2e.load_scaler = lambda path: sc1 if "latent_vectors" in path else sc2
    torch.randn_like = randn_like
    F.dropout = keep_all_dropout
    config = dict(LC_alpha=2.0, use_latent_regularization=1, latent_reg_weight=0.5, e2e_loss_function="Huber")
    try:
        with Recorder("MSELoss", "HuberLoss") as rec:
            ret, lines = run_in_tmp(lambda: ref_e2e.train_latent_conditioner_e2e(EPOCHS, train, val, m, 1e-4, 1e-4, True, IMG, config))
    finally:
        ref_e2e.load_vae_model, ref_e2e.load_scaler, torch.randn_like, F.dropout = real
        RefLC.apply = orig_apply
    epochs = parse(lines, r"Train: ([0-9.E+-]+) \(recon:([0-9.E+-]+), reg:([0-9.E+-]+)\), Val: ([0-9.E+-]+) \(recon:([0-9.E+-]+), reg:([0-9.E+-]+)\), "
                          r"LR: ([0-9.E+-]+) .*AvgGrad: ([0-9.E+-]+), Best: ([0-9.E+-]+)")
    assert len(epochs) == EPOCHS, lines
    out = dict(meta=np.array([LATENT_END, LATENT, SIZE2, IMG, B, P_TRAIN, P_VAL, EPOCHS, STATE_SEED, DATA_SEED, NOISE_SEED, calls["k"]], dtype=np.int64),
               filters=np.array(FILTERS), lr0=np.array(1e-4), wd=np.array(1e-4), ret=np.array(ret),
               huber=np.array([v for n, v in rec.values if n == "HuberLoss"]), mse=np.array([v for n, v in rec.values if n == "MSELoss"]),
               epochs=epochs)
    final_state(m, out)
    np.savez_compressed(os.path.join(HERE, "loop_e2e.npz"), **out)
    print("loop_e2e: randn_like calls", calls["k"], "huber", out["huber"], "\n", epochs)


if __name__ == "__main__":
    torch.set_num_threads(8)
    plain_loop()
    e2e_loop()
