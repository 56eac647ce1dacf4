"""Full-width / full-size golden fixtures, produced by running the REFERENCE itself on CPU.

Run only in the build container (needs /root/reference):

    python tests/golden/gen_fixtures_big.py            # g2 (seconds) and g3 (about a minute, ~12 GB of RAM)

  G2  preset filters [1024, 512, 256, 128], N = 4096, T = 32, B = 4  (BASELINE.json configs[0]'s shape; 247.5 M parameters)
  G3  preset filters, N = 95008, T = 200, B = 2                      (configs[1]'s full size; 438.2 M parameters)

The models are far too big to ship, so the weights are regenerated on any box from numpy Philox streams
(simulgen_vae_amd.init.init_state) and the fixtures hold only reference OUTPUTS of one training step: the five scalars,
the loss, the gradient norm, the 2-norm of every gradient tensor, and a few thousand sampled activation / gradient
elements (fixed, seeded positions).  tests/test_bigfix_gpu.py replays the step on the fp32 and bf16 engines.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_fixtures as gf  # noqa: E402  (registers the stand-in modules, imports the reference)
from simulgen_vae_amd.init import synthetic_eps, synthetic_samples  # noqa: E402

ALPHA, BETA = gf.ALPHA, gf.BETA
NSAMP = 96      # sampled elements per tensor
CONFIGS = {
    "g2_preset_4096": dict(latent_dim=32, hierarchical_dim=8, num_filter_enc=[1024, 512, 256, 128], num_node=4096, num_time=32, batch=4),
    "g3_fullsize_b2": dict(latent_dim=32, hierarchical_dim=8, num_filter_enc=[1024, 512, 256, 128], num_node=95008, num_time=200, batch=2),
}


def sample_positions(name, numel, n=NSAMP):
    """Fixed positions into the flattened tensor `name`: the test regenerates them from the name alone."""
    seed = int.from_bytes(name.encode()[-8:].rjust(8, b"\0"), "little") % (2 ** 31)
    rng = np.random.Generator(np.random.Philox(key=[977, seed]))
    return rng.integers(0, numel, size=min(n, numel))


def run(tag, cfgd):
    cfg, model = gf.build(cfgd, True, "MSE")
    B = cfgd["batch"]
    inj = gf.EpsInjector()
    torch.randn_like = inj
    acts = {}

    def hook(name):
        def f(mod, inp, o):
            acts[name] = o.detach()
        return f

    hs = []
    for i, blk in enumerate(model.encoder.encoder_residual_blocks):
        hs.append(blk.register_forward_hook(hook(f"enc_h{i}")))
    for i, blk in enumerate(model.decoder.decoder_residual_blocks):
        hs.append(blk.register_forward_hook(hook(f"dec_out{i}")))
    model.train(True)
    x = torch.from_numpy(synthetic_samples(gf.DATA_SEED, range(B), cfg.num_node, cfg.num_time))
    eps = synthetic_eps(gf.EPS_SEED, 0, cfg, B)
    inj.queue = [torch.from_numpy(e) for e in eps]
    xhat, recon, kls, mse = model(x)
    assert not inj.queue
    loss = recon * ALPHA + sum(kls) * BETA
    loss.backward()
    out = {}
    tot = 0.0
    nograd = []
    for n, p in model.named_parameters():
        if p.grad is None:
            nograd.append(n)
            continue
        g = p.grad.detach()
        nrm = float(g.double().norm().item())
        tot += nrm ** 2
        out["gradnorm." + n] = np.float64(nrm)
        pos = sample_positions(n, g.numel())
        out["gradsamp." + n] = g.reshape(-1)[torch.from_numpy(pos)].numpy().copy()
    out["nograd"] = np.array(nograd)
    out["scalars0"] = np.array([recon.item()] + [k.item() for k in kls] + [mse.item(), loss.item(), tot ** 0.5], dtype=np.float64)
    acts["x_hat"] = xhat.detach()
    for k, v in acts.items():
        pos = sample_positions(k, v.numel(), 512)
        out["actsamp." + k] = v.reshape(-1)[torch.from_numpy(pos)].numpy().copy()
        out["actnorm." + k] = np.float64(v.double().norm().item())
    sd = model.state_dict()
    for k in sd:            # power iteration moved u / v: norms + samples of the big ones
        if k.endswith("weight_u") or k.endswith("weight_v"):
            pos = sample_positions(k, sd[k].numel(), 32)
            out["uv1samp." + k] = sd[k].reshape(-1)[torch.from_numpy(pos)].numpy().copy()
    for h in hs:
        h.remove()
    torch.randn_like = gf._REAL_RANDN_LIKE
    out["meta"] = np.array([ALPHA, BETA, gf.STATE_SEED, gf.DATA_SEED, gf.EPS_SEED, B, cfg.num_node, cfg.num_time], dtype=np.float64)
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **out)
    print(f"{tag}: {os.path.getsize(path) / 1e3:.1f} kB, scalars0 {out['scalars0']}", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or list(CONFIGS)
    for tag in which:
        run(tag, CONFIGS[tag])
