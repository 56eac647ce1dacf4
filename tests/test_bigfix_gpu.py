"""Full-width and full-size parity against the REFERENCE (fixtures: tests/golden/gen_fixtures_big.py,
which imports /root/reference in the build container and records one training step of the reference model).

  g2_preset_4096   preset filters [1024, 512, 256, 128], N = 4096, T = 32, B = 4   (BASELINE.json configs[0]'s shape)
  g3_fullsize_b2   preset filters, N = 95008, T = 200, B = 2                       (configs[1]'s full size)

Weights / inputs / noise are regenerated here from the same numpy Philox streams, so the engine runs the very step the
reference ran.  Stated tolerances:
  fp32 engine: scalars 2e-5, per-tensor gradient norms 2e-4, sampled activations / gradients 2e-4 (of the tensor's max sample)
  bf16 engine (the bench dtype): ELBO (alpha*recon + beta*sum KL) and the reconstruction terms within 1e-4 relative of the
  reference -- the north-star bound -- KL terms 5e-3, gradient norms 3e-2, gradient norm total 1e-2, sampled activations 3e-2 of the tensor's scale.
"""
import os

import numpy as np
import pytest
import torch

import simulgen_vae_amd  # noqa: F401
from simulgen_vae_amd import engine as E
from simulgen_vae_amd.init import init_state, synthetic_eps, synthetic_samples
from simulgen_vae_amd.spec import VAEConfig

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ENC = [1024, 512, 256, 128]


def sample_positions(name, numel, n=96):
    """Same positions as tests/golden/gen_fixtures_big.py::sample_positions."""
    seed = int.from_bytes(name.encode()[-8:].rjust(8, b"\0"), "little") % (2 ** 31)
    rng = np.random.Generator(np.random.Philox(key=[977, seed]))
    return rng.integers(0, numel, size=min(n, numel))


def _run(tag, dtype):
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    alpha, beta, sseed, dseed, eseed, B, N, T = g["meta"]
    B, N, T = int(B), int(N), int(T)
    cfg = VAEConfig(32, 8, ENC, ENC[::-1], N, T, "MSE", True)
    state = init_state(cfg, int(sseed))
    eng = E.Engine(cfg, max_batch=B, compute_dtype=dtype)
    eng.load_state(state)
    x = synthetic_samples(int(dseed), range(B), N, T)
    eps = synthetic_eps(int(eseed), 0, cfg, B)
    eng.set_input(torch.from_numpy(x).cuda())
    eng.set_eps([torch.from_numpy(e).cuda() for e in eps])
    sc = eng.forward(train=True)
    acts = {}
    for i, c in enumerate(cfg.num_filter_enc):
        acts[f"enc_h{i}"] = eng.activation(f"enc_h{i}", (B, c, T))
    for i in range(len(cfg.num_filter_dec) - 1):
        acts[f"dec_out{i}"] = eng.activation(f"dec_out{i}", (B, cfg.num_filter_dec[i + 1], T))
    acts["x_hat"] = eng.activation("x_hat", (B, N, T))
    eng.backward(float(alpha), float(beta))
    gn = eng.grad_norm()
    return g, cfg, eng, sc, acts, gn, float(alpha), float(beta)


# (per-tensor gradient norm, sampled entry / scale) against the reference's gradients: measured worst case per fixture x 2-3
# (round 3, MI355X: f32 6.4e-7 / 3.9e-6 and 4.1e-6 / 9.7e-6; bf16 4.3e-3 / 4.1e-2 and 3.7e-3 / 2.3e-2)
GRAD_BOUNDS = {("g2_preset_4096", "f32"): (5e-6, 2e-5), ("g3_fullsize_b2", "f32"): (2e-5, 4e-5),
               ("g2_preset_4096", "bf16"): (1e-2, 8e-2), ("g3_fullsize_b2", "bf16"): (1e-2, 6e-2)}


@pytest.mark.parametrize("tag", ["g2_preset_4096", "g3_fullsize_b2"])
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_engine_matches_reference_fullwidth(tag, dtype):
    g, cfg, eng, sc, acts, gn, alpha, beta = _run(tag, dtype)
    ref = g["scalars0"]            # recon, kl, kl2_0, kl2_1, mse, loss, grad norm
    got = np.array([sc["recon"]] + list(sc["kls"]) + [sc["mse"]])
    elbo = alpha * sc["recon"] + beta * sum(sc["kls"])
    elbo_rel = abs(elbo - ref[5]) / abs(ref[5])
    print(f"[{tag} {dtype}] ELBO {elbo:.8e} vs reference {ref[5]:.8e}: rel {elbo_rel:.2e}; grad norm {gn:.6e} vs {ref[6]:.6e}")
    f32 = dtype == "f32"
    np.testing.assert_allclose(got[[0, 4]], ref[[0, 4]], rtol=2e-5 if f32 else 1e-4)        # reconstruction terms
    np.testing.assert_allclose(got[1:4], ref[1:4], rtol=2e-5 if f32 else 5e-3)              # KL terms (beta = 1e-4 of the ELBO)
    assert elbo_rel < (2e-5 if f32 else 1e-4), elbo_rel            # north star: ELBO within 1e-4 relative of the reference
    assert abs(gn - ref[6]) <= (2e-4 if f32 else 1e-2) * ref[6]
    nograd = set(g["nograd"].tolist())
    worst, worst_name, worst_norm, worst_norm_name = 0.0, "", 0.0, ""
    for k in g.files:
        if k.startswith("gradnorm."):
            name = k[9:]
            eg = eng.grad(name)
            assert eg is not None, name
            n2 = float(np.linalg.norm(eg.astype(np.float64)))
            rn = abs(n2 - float(g[k])) / (float(g[k]) + 1e-30)
            if rn > worst_norm:
                worst_norm, worst_norm_name = rn, name
            samp = g["gradsamp." + name]
            pos = sample_positions(name, eg.size)
            d = np.abs(eg.reshape(-1)[pos].astype(np.float64) - samp).max()
            scale = max(float(np.abs(samp).max()), float(g[k]) / np.sqrt(eg.size))
            if d / scale > worst:
                worst, worst_name = d / scale, name
    print(f"[{tag} {dtype}] worst per-tensor gradient norm error {worst_norm:.3e} ({worst_norm_name}); worst sampled-entry error / scale "
          f"{worst:.3e} ({worst_name})")
    # bounds = what the runs show plus margin (round 3: printed above, GRAD_BOUNDS below), not a generic bf16 allowance: a wrong tap
    # or a dropped 1 % term in a bf16-only kernel path moves a sampled entry by far more than this
    nb, sb = GRAD_BOUNDS[(tag, dtype)]
    assert worst_norm <= nb, (worst_norm_name, worst_norm)
    assert worst <= sb, (worst_name, worst)
    for name in nograd:
        assert eng.grad(name) is None, name
    for k in g.files:
        if k.startswith("actsamp."):
            name = k[8:]
            a = acts[name]
            pos = sample_positions(name, a.size, 512)
            d = np.abs(a.reshape(-1)[pos].astype(np.float64) - g[k]).max()
            scale = float(g["actnorm." + name]) / np.sqrt(a.size)
            assert d <= (2e-4 if f32 else 3e-2) * max(scale, float(np.abs(g[k]).max())), (name, d, scale)
    sd = None
    for k in g.files:
        if k.startswith("uv1samp."):
            if sd is None:
                sd = eng.state_dict()
            name = k[8:]
            pos = sample_positions(name, sd[name].size, 32)
            d = np.abs(sd[name].reshape(-1)[pos].astype(np.float64) - g[k]).max()
            assert d <= (1e-5 if f32 else 2e-3) * max(float(np.abs(g[k]).max()), 1e-3), (name, d)
    eng.close()
