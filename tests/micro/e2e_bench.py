"""One batch of the end-to-end conditioner -> frozen decoder loop at preset-1 size, timed in pieces.
   python tests/micro/e2e_bench.py [image_side=256] [steps=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import simulgen_vae_amd
from simulgen_vae_amd import ops
simulgen_vae_amd.install_reference_api()
from modules.VAE_network import VAE
from modules.latent_conditioner_model_cnn import LatentConditionerImg
from modules.latent_conditioner import LCOptimizer
from modules import latent_conditioner_e2e as e2e
from sklearn.preprocessing import MinMaxScaler

side = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, N, T, LAT, HIER = 16, 95008, 200, 32, 8
enc = [1024, 512, 256, 128]
vae = VAE(LAT, HIER, enc, enc[::-1], N, T, lossfun="MSE", batch_size=B, small=True).eval()
lc = LatentConditionerImg([32, 64, 128, 256, 512, 1024], LAT, (1, side, side), HIER, 3, (side, side), dropout_rate=0.2).train()
opt = LCOptimizer(lc, 1e-3, 1e-5)
rng = np.random.default_rng(0)
sc1 = MinMaxScaler((-0.7, 0.7)).fit(rng.standard_normal((64, LAT)))
sc2 = MinMaxScaler((-0.7, 0.7)).fit(rng.standard_normal((64, 3 * HIER)))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand((B, side * side), generator=g, device="cuda")
y1 = torch.randn((B, LAT), generator=g, device="cuda") * 0.3
y2 = torch.randn((B, 3, HIER), generator=g, device="cuda") * 0.3
target = torch.rand((B, N, T), generator=g, device="cuda") * 1.4 - 0.7
marks = {}


def tick(name, t0):
    torch.cuda.synchronize()
    marks[name] = marks.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()


def step(record):
    t = time.perf_counter()
    xa, ta, y1a, y2a = e2e.data_augmentation(x, target, y1, y2, True, "cuda", True)
    if record: t = tick("augmentation (4 noise passes, 1.2 GB target)", t)
    p1, p2 = lc(xa)
    if record: t = tick("conditioner forward", t)
    d1, d2 = e2e.descale_latent_predictions(p1, p2, sc1, sc2)
    rec, _ = vae.decoder(d1, e2e._decoder_latents(d2))
    if record: t = tick("descale + decoder inference (incl. [B,N,T] fp32 output)", t)
    val = float(ops.loss_value("MSE", rec, ta))
    if record: t = tick("reconstruction loss value", t)
    lc.loss_backward(xa, y1a, y2a, w1=0.0009, w2=0.0001, preds=(p1, p2))
    opt.clip_and_step(10.0, 1e-3)
    if record: t = tick("conditioner backward + clip + AdamW", t)
    return val


for _ in range(2):
    step(False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    v = step(False)
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / steps
for _ in range(steps):
    step(True)
print(f"e2e step at {side}x{side}, batch {B}, preset-1 decoder: {el*1e3:.1f} ms/step = {B/el:.0f} samples/s (recon loss {v:.4f})")
for k, s in marks.items():
    print(f"   {s/steps*1e3:7.2f} ms  {k}")
