"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

PyTorch-CPU functional restatement (fp32, autograd for the backward) of the image latent conditioner of
leesihun/SimulGen-VAE, from explicit parameters in the reference's state_dict layout:

  LatentConditionerImg.forward      reference modules/latent_conditioner_model_cnn.py:293-362
  ResNetBlock / SqueezeExcitation   reference modules/latent_conditioner_model_cnn.py:28-135
  legacy spectral norm              torch nn/utils/spectral_norm.py via modules/common.py:15-37
  loss / clip / AdamW of the loop   reference modules/latent_conditioner.py:285-314

Parity pin: tests/golden/lc_small.npz was recorded from the reference model itself (tests/golden/gen_lc_fixtures.py);
tests/test_oracle_golden.py checks this file against it.  Only tests/ and bench.py's cpu_baseline leg may import it.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def num_groups(c):
    for g in (32, 16, 8, 4, 2, 1):
        if c % g == 0 and g <= c:
            return g
    return 1


class TorchPortLC:
    """state: dict name -> tensor/array with the reference's keys (weight_orig / weight_u / weight_v, running stats)."""

    def __init__(self, filters, latent_dim_end, latent_dim, size2, state, dropout_rate=0.3, use_attention=True):
        self.filters = [int(v) for v in filters]
        self.latent_dim_end, self.latent_dim, self.size2 = latent_dim_end, latent_dim, size2
        self.r, self.use_attention = dropout_rate, use_attention
        self.S = {k: torch.as_tensor(v).clone().float() if not str(k).endswith("num_batches_tracked") else torch.as_tensor(v).clone()
                  for k, v in state.items()}
        self.trainable = [k for k in self.S if not (k.endswith("weight_u") or k.endswith("weight_v") or "running_" in k or k.endswith("num_batches_tracked"))]
        for k in self.trainable:
            self.S[k].requires_grad_()
        self.training = True
        self.adam = {}
        self.steps = 0

    def _weff(self, prefix):
        W = self.S[prefix + ".weight_orig"]
        Wm = W.view(W.shape[0], -1)
        u, v = self.S[prefix + ".weight_u"], self.S[prefix + ".weight_v"]
        if self.training:
            with torch.no_grad():
                v.copy_(F.normalize(Wm.t() @ u, dim=0, eps=1e-12))
                u.copy_(F.normalize(Wm @ v, dim=0, eps=1e-12))
        return W / torch.dot(u, Wm @ v)

    def _gn(self, prefix, x):
        return F.group_norm(x, num_groups(x.shape[1]), self.S[prefix + ".weight"], self.S[prefix + ".bias"], 1e-5)

    def _bn(self, prefix, x):
        return F.batch_norm(x, self.S[prefix + ".running_mean"], self.S[prefix + ".running_var"], self.S[prefix + ".weight"],
                            self.S[prefix + ".bias"], self.training, 0.1, 1e-5)

    def forward(self, x, masks=None):
        S, r = self.S, self.r
        B = x.shape[0]
        side = int(math.sqrt(x.shape[-1]))
        x = x.reshape(B, 1, side, side).float()
        if x.min() < -0.1:
            x = (x + 1) / 2
        masks = list(masks) if masks is not None else None

        def drop(t, p):
            if not self.training or p == 0.0:
                return t
            m = masks.pop(0) if masks is not None else (torch.rand(t.shape) >= p).float()
            return t * m / (1 - p)
        h = F.max_pool2d(F.relu(self._gn("initial_conv.1", F.conv2d(x, self._weff("initial_conv.0"), None, 1, 3))), 3, 2, 1)
        cin = self.filters[0]
        for i, cout in enumerate(self.filters[1:]):
            p = f"layers.{i}"
            stride = 2 if i in (1, 3) else 1
            o = F.relu(self._gn(p + ".gn1", F.conv2d(h, self._weff(p + ".conv1"))))
            o = self._gn(p + ".gn2", F.conv2d(o, self._weff(p + ".conv2"), None, stride, 1))
            if self.use_attention and 2 <= i <= 4:
                y = o.mean(dim=(2, 3))
                y = F.relu(F.linear(y, S[p + ".se.fc1.weight"], S[p + ".se.fc1.bias"]))
                y = torch.sigmoid(F.linear(y, S[p + ".se.fc2.weight"], S[p + ".se.fc2.bias"]))
                o = o * y[:, :, None, None]
            sk = self._gn(p + ".skip.1", F.conv2d(h, self._weff(p + ".skip.0"), None, stride)) if (stride != 1 or cin != cout) else h
            h = F.relu(o + sk)
            cin = cout
        f = drop(h.mean(dim=(2, 3)), r * 0.3)
        hid = f.shape[1] * 2
        f = F.relu(F.layer_norm(F.linear(f, self._weff("feature_processor.1"), S["feature_processor.1.bias"]), (hid,),
                                S["feature_processor.2.weight"], S["feature_processor.2.bias"]))
        f = drop(f, r * 0.4)
        f = F.relu(F.layer_norm(F.linear(f, self._weff("feature_processor.5"), S["feature_processor.5.bias"]), (hid,),
                                S["feature_processor.6.weight"], S["feature_processor.6.bias"]))
        features = drop(f, r * 0.4)

        def head(name, skipn, outn):
            t = F.linear(features, self._weff(name + "_layer1.0"), S[name + "_layer1.0.bias"])
            t = drop(F.relu(self._bn(name + "_layer1.1", t)), r * 0.3)
            t = F.linear(t, self._weff(name + "_layer2.0"), S[name + "_layer2.0.bias"])
            t = drop(F.relu(self._bn(name + "_layer2.1", t)), 0.2)
            return F.linear(t + F.linear(features, S[skipn + ".weight"], S[skipn + ".bias"]), S[outn + ".weight"], S[outn + ".bias"])
        main = head("latent_main", "main_skip_proj", "latent_main_output")
        xs = head("xs", "xs_skip_proj", "xs_output")
        return main, xs.view(B, self.size2, self.latent_dim)

    def loss_backward(self, x, y1, y2, masks=None):
        for k in self.trainable:
            self.S[k].grad = None
        p1, p2 = self.forward(x, masks)
        A, Bl = F.mse_loss(p1, torch.as_tensor(y1).float()), F.mse_loss(p2, torch.as_tensor(y2).float())
        loss = 10 * A + Bl
        loss.backward()
        return float(loss.detach()), float(A.detach()), float(Bl.detach()), p1.detach(), p2.detach()

    def clip_and_step(self, lr, weight_decay, max_norm=10.0):
        params = [self.S[k] for k in self.trainable if self.S[k].grad is not None]
        total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)).float()
        coef = min(1.0, max_norm / (float(total) + 1e-6))
        self.steps += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        with torch.no_grad():
            for k in self.trainable:
                p = self.S[k]
                if p.grad is None:
                    continue
                g = p.grad * coef
                m, v = self.adam.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
                p.mul_(1 - lr * weight_decay)
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(b2).addcmul_(g, g, value=1 - b2)
                p.addcdiv_(m, (v.sqrt() / math.sqrt(1 - b2 ** self.steps)).add_(eps), value=-lr / (1 - b1 ** self.steps))
        return float(total)
