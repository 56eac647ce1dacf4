"""SURVEY 8(f) N4 on the GPU: the end-to-end conditioner -> frozen decoder loop and ReconstructionEvaluator against an
oracle COMPOSED of parts that are each pinned by a reference-recorded fixture (conditioner: oracle/lc_torch_port.py <-
tests/golden/lc_small.npz; decoder: oracle/torch_port.py <- tests/golden/g0_*.npz; scalers: sklearn's own MinMaxScaler;
losses: torch.nn.functional).  The loop as a whole is pinned separately against a run of the reference's own loop
(tests/test_lc_loop_gpu.py <- tests/golden/loop_e2e.npz); this file covers the cases that fixture does not hold (other loss
kinds, the evaluator, error paths) against the composed oracle."""
import math
import os
import pickle

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import simulgen_vae_amd
from simulgen_vae_amd import ops
from simulgen_vae_amd.init import synthetic_samples
from simulgen_vae_amd.spec import param_spec
from tests.gpu_common import G0, GOLD, make_cfg, relerr

simulgen_vae_amd.install_reference_api()
from modules.VAE_network import VAE  # noqa: E402
from modules import latent_conditioner_e2e as e2e  # noqa: E402
from modules import utils as U  # noqa: E402
from modules.data_preprocess import latent_conditioner_scaler  # noqa: E402
from modules.latent_conditioner_model_cnn import LatentConditionerImg  # noqa: E402
from modules.reconstruction_evaluator import ReconstructionEvaluator  # noqa: E402

pytestmark = pytest.mark.gpu
LC_FILTERS = [16, 32, 32, 64, 64, 128]          # the conditioner of tests/golden/lc_small.npz; heads: hidden 256 -> 64


def _dropout_keep_all(self, x, p, masks):
    """nn.Dropout with a mask of ones (kept 1/(1-p) scaling): what the oracle gets from masks=[ones, ones]."""
    if not self.training or p == 0.0:
        return x, (lambda d: d)
    mask, scale = torch.ones_like(x), 1.0 / (1.0 - p)
    return ops.mask_scale(x, mask, scale), (lambda d: ops.mask_scale(d, mask, scale))


@pytest.mark.parametrize("kind,delta", [("MSE", 0.0), ("MAE", 0.0), ("Huber", 0.1), ("SmoothL1", 0.1), ("Huber", 0.7)])
@pytest.mark.parametrize("n", [4096 * 33 + 3, 1001, 8])
def test_loss_value_matches_torch(kind, delta, n):
    g = torch.Generator().manual_seed(5)
    a, b = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.5
    ref = dict(MSE=lambda: F.mse_loss(a.double(), b.double()), MAE=lambda: F.l1_loss(a.double(), b.double()),
               Huber=lambda: F.huber_loss(a.double(), b.double(), delta=delta),
               SmoothL1=lambda: F.smooth_l1_loss(a.double(), b.double(), beta=delta))[kind]()
    got = float(ops.loss_value(kind, a.cuda(), b.cuda(), delta if delta else 0.1))
    assert abs(got - float(ref)) < 2e-6 * abs(float(ref)), (kind, n, got, float(ref))


def test_descale_matches_sklearn_inverse_transform(tmp_path):
    rng = np.random.default_rng(2)
    lat, xs = rng.standard_normal((9, 32)) * 3 + 1, rng.standard_normal((9, 3, 8)) * 0.2
    s_lat, sc1 = latent_conditioner_scaler(lat, str(tmp_path / "latent_vectors_scaler.pkl"))
    s_xs, sc2 = latent_conditioner_scaler(xs.reshape(9, -1), str(tmp_path / "xs_scaler.pkl"))
    assert s_lat.min() >= -0.7 - 1e-9 and s_lat.max() <= 0.7 + 1e-9
    sc1b = e2e.load_scaler(str(tmp_path / "latent_vectors_scaler.pkl"))
    np.testing.assert_allclose(sc1b.min_, sc1.min_)
    p1 = rng.uniform(-0.9, 0.9, (4, 32)).astype(np.float32)
    p2 = rng.uniform(-0.9, 0.9, (4, 3, 8)).astype(np.float32)
    d1, d2 = e2e.descale_latent_predictions(torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda(), sc1, sc2)
    assert d2.shape == (4, 3, 8)
    assert relerr(d1.cpu().numpy(), sc1.inverse_transform(p1)) < 2e-6
    assert relerr(d2.cpu().numpy().reshape(4, -1), sc2.inverse_transform(p2.reshape(4, -1))) < 2e-6
    assert e2e.descale_latent_predictions(1, 2, None, sc2) == (1, 2)


def _setup(tmp_path, monkeypatch, B):
    """Small VAE (golden state g0), conditioner (random state), fitted latent scalers, 12-sample E2E dataset."""
    from oracle.lc_torch_port import TorchPortLC
    from oracle.torch_port import TorchPortVAE
    monkeypatch.chdir(tmp_path)
    os.makedirs("model_save", exist_ok=True)
    g = np.load(os.path.join(GOLD, "g0_small_MSE.npz"))
    cfg = make_cfg(G0)
    state = {e.name: g["s3." + e.name] for e in param_spec(cfg)}
    vae = VAE(cfg.latent_dim, cfg.hierarchical_dim, cfg.num_filter_enc, cfg.num_filter_dec, cfg.num_node, cfg.num_time,
              lossfun="MSE", batch_size=B, small=True, compute_dtype="f32")
    vae.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    vae.eval()
    rng = np.random.default_rng(11)
    eps_maps = [rng.standard_normal((B, c, cfg.num_time)).astype(np.float32) for c in cfg.num_filter_dec[1:-1]]
    plain_decode = vae._decode

    def decode_with_fixed_noise(z, xs=None, mode="random", freeze_level=-1):
        n = z.shape[0]       # injected noise is consumed by ONE engine forward (sgv_set_eps), so inject before every decode
        vae._eng(n).set_eps([torch.zeros(n, cfg.latent_dim).cuda()] + [torch.from_numpy(e[:n]).cuda() for e in eps_maps])
        return plain_decode(z, xs, mode)
    vae.decoder = decode_with_fixed_noise
    P, img = 12, 16
    lat = rng.standard_normal((P, cfg.latent_dim)) * 2.0
    xs = rng.standard_normal((P, 3, cfg.hierarchical_dim)) * 0.5
    s_lat, sc1 = latent_conditioner_scaler(lat, "./model_save/latent_vectors_scaler.pkl")
    s_xs, sc2 = latent_conditioner_scaler(xs.reshape(P, -1), "./model_save/xs_scaler.pkl")
    cond = rng.uniform(0, 1, (P, img * img)).astype(np.float32)
    target = synthetic_samples(5, range(P), cfg.num_node, cfg.num_time)
    ds = U.E2ELatentConditionerDataset(cond, np.float32(s_lat), np.float32(s_xs.reshape(P, 3, -1)), target, load_all=True)
    lc = LatentConditionerImg(LC_FILTERS, cfg.latent_dim, (1, img, img), cfg.hierarchical_dim, 3, (img, img), dropout_rate=0.0,
                              use_attention=True, compute_dtype="f32")
    orc_lc = TorchPortLC(LC_FILTERS, cfg.latent_dim, cfg.hierarchical_dim, 3, {k: v.clone() for k, v in lc.state_dict().items()},
                         dropout_rate=0.0, use_attention=True)
    orc_vae = TorchPortVAE(cfg, state)
    return cfg, vae, eps_maps, ds, lc, orc_lc, orc_vae, sc1, sc2


def _oracle_recon(orc_vae, eps_maps, sc1, sc2, p1, p2, mode="random"):
    d1 = sc1.inverse_transform(p1.detach().numpy())
    d2 = sc2.inverse_transform(p2.detach().numpy().reshape(p2.shape[0], -1)).reshape(p2.shape)
    with torch.no_grad():
        orc_vae._W = {}
        orc_vae.training = False
        xh, _ = orc_vae.decoder(torch.from_numpy(d1).float(), [torch.from_numpy(d2[:, i]).float() for i in range(3)],
                                [torch.from_numpy(e)[:p1.shape[0]] for e in eps_maps], mode=mode)
    return xh


def test_e2e_training_loop_matches_composed_oracle(tmp_path, monkeypatch, capsys):
    B = 4      # BatchNorm1d heads: at batch 2 the normalised values are +-1 and the gradient through them is pure cancellation noise
    cfg, vae, eps_maps, ds, lc, orc_lc, orc_vae, sc1, sc2 = _setup(tmp_path, monkeypatch, B)
    train = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, list(range(8))), batch_size=B, shuffle=False, drop_last=True)
    val = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, [8, 9, 10, 11]), batch_size=B, shuffle=False)
    monkeypatch.setattr(e2e, "load_vae_model", lambda path, device=None: vae)      # the prepared model (fixed decoder noise)
    monkeypatch.setattr(e2e, "_add_noise", lambda t, std: t)                       # augmentation noise is torch.randn on the device
    monkeypatch.setattr(LatentConditionerImg, "apply", lambda self, fn: self)      # re-initialisation tested separately
    monkeypatch.setattr(LatentConditionerImg, "_dropout", _dropout_keep_all)       # the two fixed p = 0.2 head dropouts
    ones = lambda: [torch.ones(B, 64), torch.ones(B, 64)]
    config = dict(LC_alpha=2.0, use_latent_regularization=1, latent_reg_weight=0.5, e2e_loss_function="Huber")
    lr0, wd, epochs = 1e-5, 1e-4, 2      # small steps: Adam's first steps turn rounding noise on near-zero gradients into +-lr
    s0 = lc.state_dict()
    ret = e2e.train_latent_conditioner_e2e(epochs, train, val, lc, lr0, wd, True, 16, config)
    out = capsys.readouterr().out

    # ---- the same two epochs on the composed oracle ----
    reg_w = 0.5
    log = []
    for epoch in range(epochs):
        lr = 1e-8 + (lr0 - 1e-8) * (1 + math.cos(math.pi * epoch / epochs)) / 2
        tl = tr_ = tg = 0.0
        gn = []
        for bi in range(2):
            x, y1, y2, tgt = (t.cpu() for t in ds[slice(4 * bi, 4 * bi + 4)])
            for k in orc_lc.trainable:
                orc_lc.S[k].grad = None
            p1, p2 = orc_lc.forward(x, ones())
            recon = float(F.huber_loss(_oracle_recon(orc_vae, eps_maps, sc1, sc2, p1, p2), tgt, delta=0.1))
            reg = reg_w * (0.9 * F.mse_loss(p1, y1) + 0.1 * F.mse_loss(p2.reshape(-1), y2.reshape(-1)))
            reg.backward()
            tl += 2.0 * recon + float(reg.detach())
            tr_ += recon
            tg += float(reg.detach())
            gn.append(min(orc_lc.clip_and_step(lr, wd, 10.0), 10.0))
        x, y1, y2, tgt = (t.cpu() for t in ds[slice(8, 12)])
        orc_lc.training = False
        with torch.no_grad():
            p1, p2 = orc_lc.forward(x)
            vrecon = float(F.huber_loss(_oracle_recon(orc_vae, eps_maps, sc1, sc2, p1, p2), tgt, delta=0.1))
            vreg = reg_w * float(0.9 * F.mse_loss(p1, y1) + 0.1 * F.mse_loss(p2.reshape(-1), y2.reshape(-1)))
        orc_lc.training = True
        log.append((tl / 2, tr_ / 2, tg / reg_w / 2, 2.0 * vrecon + vreg, vrecon, vreg / reg_w, sum(gn) / 2))
    lines = [ln for ln in out.splitlines() if ln.startswith("[")]
    print("\n".join(lines), log)
    assert len(lines) == epochs
    for ln, ref in zip(lines, log):
        nums = [float(v) for v in __import__("re").findall(r"[-+]?\d\.\d{4}E[-+]\d+", ln)]
        # Train, recon, reg, Val, recon, reg, AvgGrad, Best
        for got, want in zip(nums[:7], ref):
            assert abs(got - want) < 2e-3 * abs(want) + 1e-9, (ln, ref)
        assert "(CosineAnnealing), RegW: 0.5000" in ln
    assert abs(ret - log[-1][3]) < 2e-3 * abs(log[-1][3]), (ret, log[-1])
    s1 = lc.state_dict()
    worst = max((relerr(s1[k].numpy(), orc_lc.S[k].detach().numpy()), k) for k in orc_lc.trainable
                if not k.endswith("_layer1.0.bias") and not k.endswith("_layer2.0.bias"))     # zero-gradient biases (see test_ops_gpu)
    assert worst[0] < 3e-3, worst
    for k in orc_lc.trainable:          # and every large tensor moved the way the oracle's did
        if s0[k].numel() >= 4096:
            da, db = (s1[k] - s0[k]).double().flatten(), (orc_lc.S[k].detach() - s0[k]).double().flatten()
            assert float(torch.dot(da, db) / (da.norm() * db.norm())) > 0.97, k
    assert os.path.exists("checkpoints/latent_conditioner_e2e_improved.pth") and os.path.exists("model_save/LatentConditioner")
    sd = torch.load("checkpoints/latent_conditioner_e2e_improved.pth", weights_only=True)
    assert set(sd) == set(s1)
    with open("model_save/LatentConditioner", "rb") as f:
        lc2 = pickle.load(f)
    lc2.eval()
    lc.eval()
    xq = ds[0][0][None]
    assert relerr(lc2(xq)[0].cpu().numpy(), lc(xq)[0].cpu().numpy()) < 1e-6

    # without the regularisation term nothing requires grad in the reference: loss.backward() raises
    with pytest.raises(RuntimeError, match="does not require grad"):
        e2e.train_latent_conditioner_e2e(1, train, val, lc, lr0, wd, True, 16, dict(LC_alpha=1.0, use_latent_regularization=0, latent_reg_weight=0.1))
    # missing scalers
    os.remove("model_save/xs_scaler.pkl")
    with pytest.raises(ValueError, match="Could not load scalers"):
        e2e.train_latent_conditioner_e2e(1, train, val, lc, lr0, wd, True, 16, config)


def test_e2e_default_path_runs_with_noise_and_reinit(tmp_path, monkeypatch, capsys):
    """The loop as shipped (device noise augmentation, init_weights re-initialisation, model loaded from model_save/):
    finite losses, plain Linear layers re-drawn with the reference's distributions, normalised layers untouched."""
    B = 2
    cfg, vae, eps_maps, ds, lc, orc_lc, orc_vae, sc1, sc2 = _setup(tmp_path, monkeypatch, B)
    torch.save(vae, "model_save/SimulGen-VAE")
    before = lc.state_dict()
    train = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, [0, 1, 2, 3]), batch_size=B, shuffle=True, drop_last=True)
    val = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, [8, 9, 10, 11]), batch_size=B, shuffle=False)
    torch.manual_seed(3)
    ret = e2e.train_latent_conditioner_e2e(1, train, val, lc, 0.0, 0.0, True, 16,
                                           dict(LC_alpha=1.0, use_latent_regularization=1, latent_reg_weight=0.001, e2e_loss_function="nope"))
    out = capsys.readouterr().out
    assert "Unknown loss function nope, using MSE" in out and "Loaded VAE model from model_save/SimulGen-VAE" in out
    assert math.isfinite(ret) and ret > 0
    after = lc.state_dict()          # lr = 0 (eta_min aside) and no weight decay: what changed is the re-initialisation
    for k in before:
        if k.endswith(".weight") and before[k].dim() == 2:
            w = after[k].numpy()
            assert not np.allclose(w, before[k].numpy()), k
            if w.shape[0] <= 64:
                assert abs(w.std() - 0.1) < 0.03 and abs(w.mean()) < 0.03, (k, w.std())
            else:
                bound = math.sqrt(6.0 / w.shape[1])
                assert np.abs(w).max() <= bound + 1e-6 and np.abs(w).max() > 0.8 * bound, k
            assert float(after[k[:-6] + "bias"].abs().max()) < 1e-6
        elif k.endswith("weight_orig"):
            assert relerr(after[k].numpy(), before[k].numpy()) < 1e-4, k


def test_reconstruction_evaluator(tmp_path, monkeypatch, capsys):
    cfg, vae, eps_maps, ds, lc, orc_lc, orc_vae, sc1, sc2 = _setup(tmp_path, monkeypatch, 1)
    lcd = U.LatentConditionerDataset(ds.condition_data.cpu().numpy(), ds.latent_main_data.cpu().numpy(), ds.latent_hier_data.cpu().numpy())
    assert len(lcd) == 12 and lcd[2][2].shape == (3, cfg.hierarchical_dim)
    original = ds.target_reconstruction_data.cpu().numpy()
    ev = ReconstructionEvaluator(vae, "cuda", cfg.num_time, debug_mode=1)
    got = ev._reconstruct_from_latents(lcd[1][1][None].cpu().numpy(), lcd[1][2][None].cpu().numpy(), sc1, sc2)
    assert got.shape == (1, cfg.num_time, cfg.num_node)
    want = _oracle_recon(orc_vae, eps_maps, sc1, sc2, lcd[1][1][None].cpu(), lcd[1][2][None].cpu(), mode="fix").numpy().swapaxes(1, 2)
    assert relerr(got, want) < 3e-4
    sub = torch.utils.data.Subset(lcd, [0, 1])
    ev.evaluate_reconstruction_comparison(lc, sub, original[:2], sc1, sc2)
    out = capsys.readouterr().out
    assert "Evaluating 2 samples..." in out and "Sample 1 Reconstruction Stats:" in out and "VAE-only MSE:" in out
    for i in range(2):
        assert os.path.getsize(f"checkpoints/reconstruction_dual_view_{i}.png") > 10000
    # the VAE+LC number printed for sample 0 is the oracle's
    lc.eval()
    orc_lc.training = False
    with torch.no_grad():
        p1, p2 = orc_lc.forward(lcd[0][0][None].cpu())
    pred = _oracle_recon(orc_vae, eps_maps, sc1, sc2, p1, p2, mode="fix").numpy().swapaxes(1, 2)
    t = int(cfg.num_time / 2)
    mse = float(np.mean((original[0][:, t] - pred[0, t, :]) ** 2))
    line = [ln for ln in out.splitlines() if "VAE+LC MSE" in ln][0]
    assert abs(float(line.split(":")[1]) - mse) < 2e-3 * mse
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(ds, [3, 4]), batch_size=1, shuffle=False)
    ev.debug_mode = 0
    ev.evaluate_reconstruction_comparison_e2e(lc, loader, original, sc1, sc2)
    assert os.path.exists("checkpoints/reconstruction_dual_view_1.png")
