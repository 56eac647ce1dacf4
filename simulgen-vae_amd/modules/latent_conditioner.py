"""Training loop of the image latent conditioner on the MI355X -- mirror of the reference's
modules/latent_conditioner.py:107-159 (`apply_outline_preserving_augmentations`), :195-211
(`setup_optimizer_and_scheduler`) and :213-386 (`train_latent_conditioner`): same arguments, per-epoch log line,
return value and files (`checkpoints/latent_conditioner.pth`, `model_save/LatentConditioner`).

What runs where: every tensor operation (model, loss, clipping, AdamW, augmentation kernels) is a HIP kernel behind
include/sgvae_ops.h; the random decisions of the augmentations are drawn on the host with the `random` / numpy generators
(the reference draws them with torch.rand on the device; same distributions, different stream) and the learning-rate
schedule is a closed form of LinearLR(0.01 -> 1, 100 epochs) followed by CosineAnnealingLR(T_max = epochs - 100,
eta_min = 1e-8).  The reference's `summary(...)`, TensorBoard writer and matplotlib debug code have no counterpart."""
from __future__ import annotations

import math
import os
import pickle
import random
import time

import numpy as np
import torch

from .. import ops


# ---- schedule --------------------------------------------------------------------------------------------------------
def lc_learning_rate(base_lr, epochs, epoch, warmup_epochs=100, eta_min=1e-8):
    """LR in effect during `epoch` (0-based) when the reference steps `warmup_scheduler` for epoch < 100 and `main_scheduler`
    afterwards (latent_conditioner.py:196-209,358-361)."""
    if epoch <= warmup_epochs:
        return base_lr * (0.01 + 0.99 * min(epoch, warmup_epochs) / warmup_epochs)
    t_max = epochs - warmup_epochs
    t = epoch - warmup_epochs
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * t / t_max)) / 2


class LCOptimizer:
    """torch.optim.AdamW(lr, weight_decay) + torch.nn.utils.clip_grad_norm_(max_norm) over the mirror's parameters."""

    def __init__(self, model, lr, weight_decay):
        self.model, self.lr, self.weight_decay = model, float(lr), float(weight_decay)
        self.state = {}
        self.steps = 0
        self.param_groups = [{"lr": self.lr}]

    def zero_grad(self, set_to_none=True):
        self.model.grads = {}

    def clip_and_step(self, max_norm=10.0, lr=None):
        """-> total gradient norm before clipping (float, as clip_grad_norm_ returns)."""
        lr = self.param_groups[0]["lr"] if lr is None else lr
        if self.model._fused():            # multi-tensor path: <G,W> dots, norm, clip coefficient, AdamW in 5 launches
            self.steps += 1
            return self.model.pset.step(lr, self.weight_decay, max_norm)
        params = self.model.named_parameters()
        acc = torch.zeros(1, dtype=torch.float64, device="cuda")
        have = [(n, p, self.model.grads[n]) for n, p in params if n in self.model.grads]
        for _, _, g in have:
            ops.sumsq(g, acc)
        coef = ops.clip_coef(acc, max_norm)
        self.steps += 1
        for n, p, g in have:
            st = self.state.get(n)
            if st is None:
                st = self.state[n] = (torch.zeros_like(p), torch.zeros_like(p))
            ops.adamw(p, g, st[0], st[1], lr, self.steps, self.weight_decay, gscale=coef)
        return float(coef[1])


def setup_optimizer_and_scheduler(latent_conditioner, latent_conditioner_lr, weight_decay, latent_conditioner_epoch):
    """Same 4-tuple shape as the reference; the two scheduler slots carry the closed-form schedule."""
    opt = LCOptimizer(latent_conditioner, latent_conditioner_lr, weight_decay)
    warmup_epochs = 100
    sched = lambda epoch: lc_learning_rate(latent_conditioner_lr, latent_conditioner_epoch, epoch, warmup_epochs)
    return opt, sched, sched, warmup_epochs


# ---- augmentation ------------------------------------------------------------------------------------------------------
def apply_outline_preserving_augmentations(x, prob=0.5, rng=random):
    """x: fp32 CUDA [B, H, W].  Same stages, probabilities and ranges as latent_conditioner.py:107-159: horizontal flip
    (stage p 0.3, per sample p 0.5), +-1 pixel roll (p 0.5), rotation by U(-5, 5) degrees when |angle| > 0.5 (p 0.3),
    scaling by U(0.95, 1.05) when |scale - 1| > 0.01 (p 0.3); bilinear resampling with border padding."""
    if not rng.random() < prob:
        return x
    B, H, W = x.shape
    flip = [0] * B
    sx, sy = [0] * B, [0] * B
    if rng.random() < 0.3:
        flip = [1 if rng.random() < 0.5 else 0 for _ in range(B)]
    if rng.random() < 0.5:
        sx = [rng.randint(-1, 1) for _ in range(B)]
        sy = [rng.randint(-1, 1) for _ in range(B)]
    if any(flip) or any(sx) or any(sy):
        x = ops.flip_roll(x, flip, sx, sy)
    ident = np.tile(np.array([[1, 0, 0], [0, 1, 0]], np.float32), (B, 1, 1))
    if rng.random() < 0.3:
        th = ident.copy()
        angles = [(rng.random() - 0.5) * 10 for _ in range(B)]
        for i, a in enumerate(angles):
            if abs(a) > 0.5:
                c, s = math.cos(a * math.pi / 180), math.sin(a * math.pi / 180)
                th[i] = [[c, -s, 0], [s, c, 0]]
        if any(abs(a) > 0.5 for a in angles):
            x = ops.affine_sample(x, th)
    if rng.random() < 0.3:
        th = ident.copy()
        scales = [0.95 + rng.random() * 0.1 for _ in range(B)]
        for i, sc in enumerate(scales):
            if abs(sc - 1.0) > 0.01:
                th[i] = [[sc, 0, 0], [0, sc, 0]]
        if any(abs(sc - 1.0) > 0.01 for sc in scales):
            x = ops.affine_sample(x, th)
    return x


def _dev(t):
    if not torch.is_tensor(t):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device="cuda", dtype=torch.float32).contiguous()


# ---- training loop -----------------------------------------------------------------------------------------------------
def train_latent_conditioner(latent_conditioner_epoch, latent_conditioner_dataloader, latent_conditioner_validation_dataloader,
                             latent_conditioner, latent_conditioner_lr, weight_decay=1e-4, is_image_data=True, rng=random):
    opt, sched, _, warmup_epochs = setup_optimizer_and_scheduler(latent_conditioner, latent_conditioner_lr, weight_decay,
                                                                 latent_conditioner_epoch)
    best_val_loss, patience, patience_counter, min_delta, overfitting_threshold = float("inf"), 100000, 0, 1e-8, 1000.0
    latent_conditioner.apply(None)          # safe_initialize_weights_He: see LatentConditionerImg.apply
    avg_val_loss = avg_val_loss_y1 = avg_val_loss_y2 = 0.0
    for epoch in range(latent_conditioner_epoch):
        start_time = time.time()
        latent_conditioner.train(True)
        lr = sched(epoch)
        opt.param_groups[0]["lr"] = lr
        epoch_loss = epoch_loss_y1 = epoch_loss_y2 = 0.0
        num_batches = 0
        for i, (x, y1, y2) in enumerate(latent_conditioner_dataloader):
            x, y1, y2 = _dev(x), _dev(y1), _dev(y2)
            if is_image_data and rng.random() < 0.5:
                side = int(math.sqrt(x.shape[-1]))
                x = apply_outline_preserving_augmentations(x.reshape(-1, side, side), prob=0.8, rng=rng).reshape(x.shape[0], -1)
            if rng.random() < 0.02 and x.shape[0] > 1:
                lam = float(np.random.beta(0.2, 0.2))
                perm = torch.randperm(x.shape[0]).tolist()
                x, y1 = ops.mixup_rows(x, perm, lam), ops.mixup_rows(y1, perm, lam)
                y2 = ops.mixup_rows(y2.reshape(y2.shape[0], -1).contiguous(), perm, lam).reshape(y2.shape)
            if rng.random() < 0.05:
                x = ops.addf(x, ops.mask_scale(torch.randn_like(x), None, 0.01))
            opt.zero_grad(set_to_none=True)
            loss, A, Bv = latent_conditioner.loss_backward(x, y1, y2)
            epoch_loss += loss
            epoch_loss_y1 += A
            epoch_loss_y2 += Bv
            num_batches += 1
            total_grad_norm = opt.clip_and_step(max_norm=10.0, lr=lr)
            if epoch % 100 == 0 and i == 0:
                print(f"DEBUG: Gradient norm: {total_grad_norm:.4f}, Loss: {loss:.4E}")
                if total_grad_norm > 10.0:
                    print(f"WARNING: Large gradient norm detected: {total_grad_norm:.2f}")
                elif total_grad_norm < 1e-4:
                    print(f"WARNING: Very small gradient norm: {total_grad_norm:.2E}")
        avg_train_loss = epoch_loss / num_batches
        avg_train_loss_y1 = epoch_loss_y1 / num_batches
        avg_train_loss_y2 = epoch_loss_y2 / num_batches
        latent_conditioner.eval()
        if epoch % 10 == 0:
            val_loss = val_loss_y1 = val_loss_y2 = 0.0
            val_batches = 0
            for x_val, y1_val, y2_val in latent_conditioner_validation_dataloader:
                p1, p2 = latent_conditioner(_dev(x_val))
                a, _ = ops.mse(p1, _dev(y1_val), need_grad=False)
                y2v = _dev(y2_val)
                b, _ = ops.mse(p2.reshape(p2.shape[0], -1).contiguous(), y2v.reshape(y2v.shape[0], -1).contiguous(), need_grad=False)
                a, b = float(a), float(b)
                val_loss += 10 * a + b
                val_loss_y1 += a
                val_loss_y2 += b
                val_batches += 1
            avg_val_loss = val_loss / val_batches
            avg_val_loss_y1 = val_loss_y1 / val_batches
            avg_val_loss_y2 = val_loss_y2 / val_batches
            overfitting_ratio = avg_val_loss / max(avg_train_loss, 1e-8)
            if overfitting_ratio > overfitting_threshold:
                print(f"Severe overfitting detected! Val/Train ratio: {overfitting_ratio:.1f}")
                print(f"Stopping early at epoch {epoch}")
                break
            if avg_val_loss < best_val_loss - min_delta:
                best_val_loss = avg_val_loss
                patience_counter = 0
            else:
                patience_counter += 1
        epoch_duration = time.time() - start_time
        current_lr = sched(epoch + 1)       # after this epoch's scheduler.step() (the reference steps it in the last epoch too)
        scheduler_info = "Warmup" if epoch < warmup_epochs else "Cosine"
        print("[%d/%d]\tTrain: %.4E (y1:%.4E, y2:%.4E), Val: %.4E (y1:%.4E, y2:%.4E), LR: %.2E (%s), ETA: %.2f h, Patience: %d/%d" %
              (epoch, latent_conditioner_epoch, avg_train_loss, avg_train_loss_y1, avg_train_loss_y2, avg_val_loss, avg_val_loss_y1,
               avg_val_loss_y2, current_lr, scheduler_info, (latent_conditioner_epoch - epoch) * epoch_duration / 3600, patience_counter, patience))
        if patience_counter >= patience:
            print(f"Early stopping at epoch {epoch}. Best validation loss: {best_val_loss:.4E}")
            break
    os.makedirs("checkpoints", exist_ok=True)
    os.makedirs("model_save", exist_ok=True)
    torch.save(latent_conditioner.state_dict(), "checkpoints/latent_conditioner.pth")
    with open("model_save/LatentConditioner", "wb") as f:
        pickle.dump(latent_conditioner, f)
    return avg_val_loss
