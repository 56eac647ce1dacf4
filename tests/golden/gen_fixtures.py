"""Generate the golden fixtures in this directory by running the REFERENCE itself on CPU.

Run only in the build container (needs /root/reference; it does not exist on the GPU box):

    python tests/golden/gen_fixtures.py

The reference is imported unmodified; four non-arithmetic helper modules that are not installed
here (torchinfo, tensorboard, skimage.util, torchvision.transforms.v2 -- SURVEY 8(c)) are
registered as empty stand-ins in sys.modules so the imports resolve.  Weights and inputs come
from simulgen_vae_amd.init (numpy Philox), so the fixtures hold only reference OUTPUTS.
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

for name, attrs in [("torchinfo", {"summary": lambda *a, **k: None}),
                    ("torch.utils.tensorboard", {"SummaryWriter": type("SummaryWriter", (), {
                        "__init__": lambda s, *a, **k: None})}),
                    ("skimage", {}), ("skimage.util", {"random_noise": None}),
                    ("torchvision", {}), ("torchvision.transforms", {}), ("torchvision.transforms.v2", {})]:
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)

import simulgen_vae_amd  # noqa: E402,F401
from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples  # noqa: E402
from simulgen_vae_amd.spec import VAEConfig, param_spec  # noqa: E402

from modules.VAE_network import VAE  # noqa: E402  (reference)
from modules.common import add_sn, initialize_weights_He  # noqa: E402
from modules.train import WarmupKLLoss  # noqa: E402
import modules.decoder as ref_decoder  # noqa: E402
import modules.VAE_network as ref_vaenet  # noqa: E402

torch.set_num_threads(8)
_REAL_RANDN_LIKE = torch.randn_like
ALPHA, BETA, LR = 1e6, 1e-4, 1e-3
STATE_SEED, DATA_SEED, EPS_SEED = 7, 20251003, 1234

CONFIGS = {
    "g0": dict(latent_dim=32, hierarchical_dim=8, num_filter_enc=[32, 16, 8, 8], num_node=72, num_time=10,
               batch=3),
    "g1": dict(latent_dim=32, hierarchical_dim=8, num_filter_enc=[64, 32, 16, 8], num_node=520, num_time=12,
               batch=4),
}


class EpsInjector:
    """Replaces torch.randn_like inside the reference's reparameterize with preset tensors."""

    def __init__(self):
        self.queue = []

    def __call__(self, t):
        e = self.queue.pop(0)
        assert tuple(e.shape) == tuple(t.shape), (e.shape, t.shape)
        return e.to(t.dtype)


def build(cfgd, small, lossfun):
    enc = cfgd["num_filter_enc"]
    cfg = VAEConfig(cfgd["latent_dim"], cfgd["hierarchical_dim"], enc, enc[::-1], cfgd["num_node"],
                    cfgd["num_time"], lossfun, small)
    model = VAE(cfg.latent_dim, cfg.hierarchical_dim, enc, enc[::-1], cfg.num_node, cfg.num_time,
                lossfun=lossfun, batch_size=cfgd["batch"], small=small)
    model.apply(initialize_weights_He)
    model.apply(add_sn)
    state = init_state(cfg, STATE_SEED)
    sd = model.state_dict()
    spec = param_spec(cfg)
    assert [e.name for e in spec] == list(sd.keys()), "param_spec order differs from reference state_dict"
    for e in spec:
        assert tuple(sd[e.name].shape) == tuple(e.shape), (e.name, sd[e.name].shape, e.shape)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    return cfg, model


def run_case(tag, cfgd, small, lossfun, full):
    cfg, model = build(cfgd, small, lossfun)
    B = cfgd["batch"]
    inj = EpsInjector()
    torch.randn_like = inj  # the reference's reparameterize() (decoder.py:221) draws through this
    out = {}
    acts = {}

    def hook(name):
        def f(mod, inp, o):
            acts[name] = o.detach().numpy().copy()
        return f

    hs = []
    for i, blk in enumerate(model.encoder.encoder_residual_blocks):
        hs.append(blk.register_forward_hook(hook(f"enc_h{i}")))
    for i, blk in enumerate(model.decoder.decoder_residual_blocks):
        hs.append(blk.register_forward_hook(hook(f"dec_out{i}")))

    opt = torch.optim.AdamW(model.parameters(), lr=LR)
    model.train(True)
    trainable = {n for n, _ in model.named_parameters()}
    steps = 3
    for step in range(steps):
        x = torch.from_numpy(synthetic_samples(DATA_SEED, range(step * B, (step + 1) * B), cfg.num_node,
                                               cfg.num_time))
        eps = synthetic_eps(EPS_SEED, step, cfg, B)
        inj.queue = [torch.from_numpy(e) for e in eps]
        opt.zero_grad(set_to_none=True)
        xhat, recon, kls, mse = model(x)
        assert not inj.queue
        loss = recon * ALPHA + sum(kls) * BETA
        loss.backward()
        tot = 0.0
        for p in model.parameters():
            if p.grad is not None:
                tot += p.grad.data.norm(2).item() ** 2
        gnorm = tot ** 0.5
        if step == 0:
            out["x_hat"] = xhat.detach().numpy()
            mu, lv, xs = None, None, None
            for k, v in acts.items():
                out["act." + k] = v
            out["scalars0"] = np.array([recon.item()] + [k.item() for k in kls] + [mse.item(), loss.item(), gnorm],
                                       dtype=np.float64)
            sd = model.state_dict()
            for k in sd:
                if k.endswith("weight_u") or k.endswith("weight_v"):
                    out["uv1." + k] = sd[k].numpy().copy()
            nograd = []
            for n, p in model.named_parameters():
                if p.grad is None:
                    nograd.append(n)
                elif full:
                    out["grad." + n] = p.grad.numpy().copy()
                else:
                    out["gradnorm." + n] = np.float64(p.grad.double().norm().item())
            out["nograd"] = np.array(nograd)
        else:
            out[f"scalars{step}"] = np.array([recon.item()] + [k.item() for k in kls] + [mse.item(), loss.item(), gnorm],
                                             dtype=np.float64)
        opt.step()
        if step == 0 and full:
            for n, p in model.named_parameters():
                if "recon.0" in n or "encoder_blocks.0" in n or "condition_xz.0.2" in n:
                    out["p1." + n] = p.detach().numpy().copy()
    sd = model.state_dict()
    for k in sd:
        if full:
            out["s3." + k] = sd[k].numpy().copy()
        else:
            out["s3norm." + k] = np.float64(sd[k].double().norm().item())
    # encoder outputs (train mode consumed a power iteration: do these in eval mode on the step-3 state)
    model.eval()
    x = torch.from_numpy(synthetic_samples(DATA_SEED, range(100, 100 + B), cfg.num_node, cfg.num_time))
    eps = synthetic_eps(EPS_SEED, 100, cfg, B)
    with torch.no_grad():
        inj.queue = [torch.from_numpy(e) for e in eps]
        xhat_e, recon_e, kls_e, mse_e = model(x)
        out["eval.x_hat"] = xhat_e.numpy()
        out["eval.scalars"] = np.array([recon_e.item()] + [k.item() for k in kls_e] + [mse_e.item()], dtype=np.float64)
        mu, lv, xs = model.encoder(x)
        out["eval.mu"], out["eval.log_var"] = mu.numpy().copy(), lv.numpy().copy()
        for i, v in enumerate(xs):
            out[f"eval.xs{i}"] = v.numpy().copy()
        # evaluate_vae_reconstruction path (utils.py:492-499): reparameterize then decoder(mode='fix')
        inj.queue = [torch.from_numpy(e) for e in eps]
        std = torch.exp(0.5 * lv)
        z = ref_decoder.reparameterize(mu, std)
        xfix, _ = model.decoder(z, xs, mode="fix")
        out["fix.x_hat"] = xfix.numpy()
        out["fix.z"] = z.numpy().copy()
    for h in hs:
        h.remove()
    torch.randn_like = _REAL_RANDN_LIKE
    out["meta"] = np.array([ALPHA, BETA, LR, STATE_SEED, DATA_SEED, EPS_SEED, B], dtype=np.float64)
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{tag}: {os.path.getsize(path) / 1e6:.2f} MB, scalars0 {out['scalars0']}")


def schedules():
    out = {}
    for E in (4, 8, 20, 40):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=LR)
        sch = torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, T_0=E // 4, T_mult=2, eta_min=LR * 0.0001)
        lrs = []
        for e in range(E):
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out[f"lr_E{E}"] = np.array(lrs, dtype=np.float64)
    for E in (4, 10, 20):
        w = WarmupKLLoss(E, 1e-4, int(E * 0.3), int(E * 0.8), 1)
        out[f"beta_E{E}"] = np.array([w.get_loss(e, [0.0])[0] for e in range(E)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "schedules.npz"), **out)
    print("schedules:", {k: v[:4] for k, v in out.items()})


def augmentation():
    """One pass of AugmentedDataset.__getitem__ over 8 samples with every random draw recorded.
    The python-random seed is searched so that every branch (noise / scale / mixup taken and not
    taken) occurs at least once."""
    import modules.augmentation as ref_aug
    N, T, P = 24, 6, 8
    data = synthetic_samples(DATA_SEED, range(P), N, T)
    ds = ref_aug.AugmentedDataset(data, load_all=True)
    real_random = ref_aug.random
    old_beta, old_randn = np.random.beta, torch.randn_like
    for seed in range(99, 400):
        log = {"rand": [], "randint": [], "beta": [], "noise": []}
        rng = random.Random(seed)
        nprng = np.random.RandomState(5)
        tg = torch.Generator().manual_seed(3)

        class R:
            @staticmethod
            def random():
                v = rng.random(); log["rand"].append(v); return v

            @staticmethod
            def randint(a, b):
                v = rng.randint(a, b); log["randint"].append(v); return v

        def beta(a, b):
            v = nprng.beta(a, b); log["beta"].append(v); return v

        def randn_like(t):
            v = torch.randn(t.shape, generator=tg); log["noise"].append(v.numpy().copy()); return v

        ref_aug.random = R
        np.random.beta = beta
        torch.randn_like = randn_like
        outs, counts = [], []
        try:
            for i in range(P):
                c0 = [len(log[k]) for k in ("rand", "randint", "beta", "noise")]
                outs.append(ds[i].numpy().copy())
                counts.append([len(log[k]) - c for k, c in zip(("rand", "randint", "beta", "noise"), c0)])
        finally:
            np.random.beta, torch.randn_like = old_beta, old_randn
            ref_aug.random = real_random
        c = np.array(counts)
        if (set(c[:, 3]) == {0, 1} and set(c[:, 2]) == {0, 1} and set(c[:, 0]) >= {5, 6}
                and any(b <= 0.1 or b >= 0.9 for b in log["beta"])):
            break
    np.savez_compressed(os.path.join(HERE, "augment.npz"), out=np.stack(outs), counts=np.array(counts),
                        rand=np.array(log["rand"]), randint=np.array(log["randint"]), beta=np.array(log["beta"]),
                        noise=np.stack(log["noise"]) if log["noise"] else np.zeros((0, N, T), np.float32),
                        shape=np.array([P, N, T]), seed=np.array([seed]))
    print("augment seed", seed, "counts:", counts, "beta", log["beta"])


def scaler():
    """Reference data_scaler / reduce_dataset (modules/data_preprocess.py:13-41,65-165) on a small synthetic raw array
    [P, T, N] with a constant node, a NaN-free heavy-tailed node and values outside the sampled range.
    P*T = 12000 rows > 10x the 1000-row sample floor, so the np.random.seed(42) row sampling really subsamples."""
    import modules.data_preprocess as dp
    rng = np.random.default_rng(123)
    P, T, N = 60, 200, 16
    raw = (rng.standard_normal((P, T, N)) * rng.uniform(0.1, 30.0, N) + rng.uniform(-5, 5, N)).astype(np.float32)
    raw[:, :, 3] = 2.5                      # zero range -> sklearn's _handle_zeros_in_scale
    raw[7, 11, 5] = 1e4                     # outlier that the row sample may or may not contain
    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        os.makedirs("model_save")
        try:
            work = raw.copy()
            out, shape, sc = dp.data_scaler(work, work, T, N, 1)
        finally:
            os.chdir(cwd)
    nt, red, nn = dp.reduce_dataset(raw.copy(), 150, 8, P, T, 4, 12)
    np.savez_compressed(os.path.join(HERE, "scaler.npz"), raw=raw, scaled=np.asarray(out, np.float32), data_min=sc.data_min_,
                        data_max=sc.data_max_, scale=sc.scale_, offset=sc.min_, shape=np.array(shape),
                        reduced=np.asarray(red, np.float32), reduced_meta=np.array([nt, nn]))
    print("scaler.npz", out.shape, float(np.min(out)), float(np.max(out)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "scaler":
        scaler()
        sys.exit(0)
    run_case("g0_small_MSE", CONFIGS["g0"], True, "MSE", full=True)
    run_case("g0_large_MSE", CONFIGS["g0"], False, "MSE", full=True)
    for lf in ("MAE", "smoothL1", "Huber"):
        run_case(f"g0_small_{lf}", CONFIGS["g0"], True, lf, full=False)
    run_case("g1_small_MSE", CONFIGS["g1"], True, "MSE", full=False)
    schedules()
    augmentation()
    scaler()
