"""End-to-end latent-conditioner training through the frozen VAE decoder on the MI355X -- mirror of the
reference's modules/latent_conditioner_e2e.py: `load_vae_model` (:32-56), `load_scaler` (:58-66),
`descale_latent_predictions` (:68-92), `setup_improved_optimizer_and_scheduler` (:120-146), `data_augmentation`
(:148-211) and `train_latent_conditioner_e2e` (:213-561): same arguments, config keys, per-epoch log line, return value
and files (`checkpoints/latent_conditioner_e2e_improved.pth`, `model_save/LatentConditioner`).

What the reference's loop computes, and so what this one does:
  * conditioner forward -> MinMaxScaler.inverse_transform of both predictions -> `vae_model.decoder(z, [xs_0, xs_1, xs_2])`
    in its default ("random") mode -> reconstruction loss against the (noised) target.  The reference descales through
    `.detach().cpu().numpy()`, so this term carries NO gradient: it is a reported value only (`ops.loss_value`), and the
    decoder runs as inference (`sgv_decode`).
  * with `use_latent_regularization = 1` the conditioner is trained by `latent_reg_weight * (0.9*MSE(y1) + 0.1*MSE(y2))`
    alone; without it the reference's `loss.backward()` fails because nothing requires grad -- the same RuntimeError is
    raised here.
  * "hybrid" clipping: norms above 10 are clipped to 10; norms below 1e-5 go through `clip_grad_norm_(params, 1e-5)`, which
    never scales up, so they are left alone (the reference's "Scaled up" message is printed all the same).
  * AdamW + CosineAnnealingLR(T_max = epochs, eta_min = 1e-8), stepped per epoch (closed form below).
Every tensor operation is a HIP kernel behind include/sgvae_ops.h / include/sgvae.h; torch supplies device memory and
the random bits of the augmentation.  TensorBoard, torchinfo and the commented-out augmentations have no counterpart."""
from __future__ import annotations

import math
import os
import pickle
import time

import numpy as np
import torch

from .. import ops
from .latent_conditioner import LCOptimizer, _dev

LOSS_FUNCTIONS = {"MSE": ("MSE", 0.0), "MAE": ("MAE", 0.0), "Huber": ("Huber", 0.1), "SmoothL1": ("SmoothL1", 0.1)}


def load_vae_model(vae_model_path, device=None):
    """Frozen, eval-mode VAE from `model_save/SimulGen-VAE` (written by modules.train.train of this package)."""
    if not os.path.exists(vae_model_path):
        print(f"VAE model not found at {vae_model_path}")
        raise FileNotFoundError(f"VAE model not found at {vae_model_path}")
    vae_model = torch.load(vae_model_path, map_location="cpu", weights_only=False)
    print(f"Loaded VAE model from {vae_model_path}")
    vae_model.eval()
    print("VAE model weights frozen for end-to-end training")
    return vae_model


def load_scaler(scaler_path):
    try:
        with open(scaler_path, "rb") as f:
            return pickle.load(f)
    except Exception as e:  # same reporting as the reference
        print(f"Error loading scaler from {scaler_path}: {e}")
        return None


class _DeviceScaler:
    """min_ / scale_ of a fitted sklearn MinMaxScaler as fp32 device vectors (inverse_transform = (y - min_) / scale_)."""

    def __init__(self, scaler):
        self.min_ = torch.as_tensor(np.asarray(scaler.min_, dtype=np.float32)).cuda()
        self.scale_ = torch.as_tensor(np.asarray(scaler.scale_, dtype=np.float32)).cuda()

    def inverse(self, y2d):
        return ops.cols_sub_div(y2d.contiguous(), self.min_, self.scale_)


def _device_scaler(scaler, cache={}):
    key = id(scaler)
    if key not in cache or cache[key][0] is not scaler:
        cache[key] = (scaler, _DeviceScaler(scaler))
    return cache[key][1]


def descale_latent_predictions(y_pred1, y_pred2, latent_vectors_scaler, xs_scaler):
    """-> (y1 descaled [B, latent_dim_end], y2 descaled, same shape as y_pred2); detached by construction."""
    if latent_vectors_scaler is None or xs_scaler is None:
        return y_pred1, y_pred2
    y1 = _device_scaler(latent_vectors_scaler).inverse(_dev(y_pred1))
    y2 = _dev(y_pred2)
    y2d = _device_scaler(xs_scaler).inverse(y2.reshape(y2.shape[0], -1)).reshape(y2.shape)
    return y1, y2d


def cosine_lr(base_lr, epochs, epoch, eta_min=1e-8):
    """LR in effect during `epoch` under CosineAnnealingLR(T_max = epochs, eta_min) stepped once per epoch."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / epochs)) / 2


def setup_improved_optimizer_and_scheduler(latent_conditioner, latent_conditioner_lr, weight_decay, latent_conditioner_epoch):
    opt = LCOptimizer(latent_conditioner, latent_conditioner_lr, weight_decay)
    return opt, (lambda epoch: cosine_lr(latent_conditioner_lr, latent_conditioner_epoch, epoch))


def _add_noise(t, std):
    return ops.addf(t, ops.mask_scale(torch.randn_like(t), None, std))


def data_augmentation(x, target_data, y1, y2, is_image_data, device, use_latent_regularization):
    """latent_conditioner_e2e.py:148-211 as it runs (the other stages are commented out there): Gaussian noise of
    std 0.1 on the input and 0.05 on the reconstruction target and both latent targets, every batch."""
    x = _add_noise(x, 0.1)
    target_data = _add_noise(target_data, 0.05)
    y1 = _add_noise(y1, 0.05)
    y2 = _add_noise(y2.reshape(y2.shape[0], -1).contiguous(), 0.05).reshape(y2.shape)
    return x, target_data, y1, y2


def _decoder_latents(y2_descaled):
    """[B, 3, d] or [B, 3*d] -> list of 3 [B, d] tensors (latent_conditioner_e2e.py:360-368)."""
    if y2_descaled.dim() == 3 and y2_descaled.shape[1] == 3:
        return [y2_descaled[:, i, :].contiguous() for i in range(3)]
    if y2_descaled.dim() == 2:
        d = y2_descaled.shape[1] // 3
        v = y2_descaled.view(y2_descaled.shape[0], 3, d)
        return [v[:, i, :].contiguous() for i in range(3)]
    return y2_descaled


def _reg_terms(p1, p2, y1, y2):
    a, _ = ops.mse(p1, y1, need_grad=False)
    b, _ = ops.mse(p2.reshape(p2.shape[0], -1).contiguous(), y2.reshape(y2.shape[0], -1).contiguous(), need_grad=False)
    return float(a), float(b)


def train_latent_conditioner_e2e(latent_conditioner_epoch, e2e_dataloader, e2e_validation_dataloader, latent_conditioner,
                                 latent_conditioner_lr, weight_decay, is_image_data, image_size, config):
    LC_alpha = float(config.get("LC_alpha")) if config else 1.0
    vae_model_path = config.get("e2e_vae_model_path", "model_save/SimulGen-VAE") if config else "model_save/SimulGen-VAE"
    vae_model = load_vae_model(vae_model_path)
    print("Loading scalers for latent prediction descaling...")
    latent_vectors_scaler = load_scaler("./model_save/latent_vectors_scaler.pkl")
    xs_scaler = load_scaler("./model_save/xs_scaler.pkl")
    if latent_vectors_scaler is None or xs_scaler is None:
        raise ValueError("Could not load scalers. E2E training will use raw latent predictions.")
    print("Scalers loaded successfully for latent prediction descaling")
    loss_function_type = config.get("e2e_loss_function", "MSE") if config else "MSE"
    if loss_function_type not in LOSS_FUNCTIONS:
        print(f"Unknown loss function {loss_function_type}, using MSE")
    loss_kind, loss_delta = LOSS_FUNCTIONS.get(loss_function_type, LOSS_FUNCTIONS["MSE"])
    use_latent_regularization = (config.get("use_latent_regularization") == 1) if config else False
    latent_reg_weight = float(config.get("latent_reg_weight")) if config else 0.001
    opt, sched = setup_improved_optimizer_and_scheduler(latent_conditioner, latent_conditioner_lr, weight_decay, latent_conditioner_epoch)

    def init_weights(m):          # latent_conditioner_e2e.py:267-285; see LatentConditionerImg.apply
        return m
    latent_conditioner.apply(init_weights)
    print(f"Starting improved end-to-end latent conditioner training for {latent_conditioner_epoch} epochs")
    print(f"Reconstruction loss function: {loss_function_type}")
    print(f"Latent regularization: {'Enabled' if use_latent_regularization else 'Disabled'}")

    def recon_value(y_pred1, y_pred2, target):
        d1, d2 = descale_latent_predictions(y_pred1, y_pred2, latent_vectors_scaler, xs_scaler)
        reconstructed, _ = vae_model.decoder(d1, _decoder_latents(d2))
        return float(ops.loss_value(loss_kind, reconstructed.contiguous(), target.contiguous(), loss_delta)), reconstructed

    best_val_loss, best_model_state, patience_counter = float("inf"), None, 0
    gradient_norms = []
    avg_val_loss = 0.0
    current_reg_weight = latent_reg_weight if use_latent_regularization else 0.0
    for epoch in range(latent_conditioner_epoch):
        epoch_start_time = time.time()
        latent_conditioner.train(True)
        lr = sched(epoch)
        opt.param_groups[0]["lr"] = lr
        epoch_loss = epoch_recon_loss = epoch_latent_reg_loss = 0.0
        num_batches = 0
        epoch_gradient_sum, gradient_count = 0.0, 0
        for i, (x, y1, y2, target_data) in enumerate(e2e_dataloader):
            x, y1, y2, target_data = _dev(x), _dev(y1), _dev(y2), _dev(target_data)
            x, target_data, y1, y2 = data_augmentation(x, target_data, y1, y2, is_image_data, "cuda", use_latent_regularization)
            opt.zero_grad(set_to_none=True)
            y_pred1, y_pred2 = latent_conditioner(x)
            if i == 0 and epoch % 100 == 0:
                print(f"Epoch {epoch} - CNN vs Target Latents:")
                print(f"   y1 - CNN: [{float(y_pred1.min()):.4f}, {float(y_pred1.max()):.4f}], Target: [{float(y1.min()):.4f}, {float(y1.max()):.4f}]")
                print(f"   y2 - CNN: [{float(y_pred2.min()):.4f}, {float(y_pred2.max()):.4f}], Target: [{float(y2.min()):.4f}, {float(y2.max()):.4f}]")
            recon_loss, reconstructed = recon_value(y_pred1, y_pred2, target_data)
            if i == 0 and epoch % 100 == 0:
                print(f"Epoch {epoch} - Reconstructed Data - range: [{float(reconstructed.min()):.4f}, {float(reconstructed.max()):.4f}]")
            if not use_latent_regularization:
                # loss = recon_loss, which the reference computed from detached numpy round-trips
                raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
            reg_loss, A, Bv = latent_conditioner.loss_backward(x, y1, y2, w1=current_reg_weight * 0.9, w2=current_reg_weight * 0.1,
                                                               preds=(y_pred1, y_pred2))
            loss = LC_alpha * recon_loss + reg_loss
            epoch_latent_reg_loss += reg_loss
            epoch_loss += loss
            epoch_recon_loss += recon_loss
            num_batches += 1
            min_grad_norm, max_grad_norm = 1e-5, 10
            original_gradient_norm = opt.clip_and_step(max_norm=max_grad_norm, lr=lr)
            if original_gradient_norm > 0:
                if original_gradient_norm < min_grad_norm:
                    final_gradient_norm = min_grad_norm
                    if i % 10 == 0:
                        print(f"  Batch {i}: Scaled up gradients by {min_grad_norm / original_gradient_norm:.2f} ({original_gradient_norm:.2E} -> {final_gradient_norm:.2E})")
                elif original_gradient_norm > max_grad_norm:
                    final_gradient_norm = max_grad_norm
                    if i % 10 == 0:
                        print(f"  Batch {i}: Scaled down gradients ({original_gradient_norm:.2E} -> {final_gradient_norm:.2E})")
                else:
                    final_gradient_norm = original_gradient_norm
            else:
                final_gradient_norm = 0.0
            gradient_norms.append(final_gradient_norm)
            epoch_gradient_sum += final_gradient_norm
            gradient_count += 1
        avg_train_loss = epoch_loss / num_batches if num_batches > 0 else 0.0
        avg_train_recon_loss = epoch_recon_loss / num_batches if num_batches > 0 else 0.0
        avg_train_latent_reg_loss = epoch_latent_reg_loss / current_reg_weight / num_batches if num_batches > 0 else 0.0

        latent_conditioner.eval()
        val_loss = val_recon_loss = val_latent_reg_loss = 0.0
        val_batches = 0
        for x_val, y1_val, y2_val, target_val_data in e2e_validation_dataloader:
            x_val, y1_val, y2_val, target_val_data = _dev(x_val), _dev(y1_val), _dev(y2_val), _dev(target_val_data)
            p1, p2 = latent_conditioner(x_val)
            recon_loss_val, _ = recon_value(p1, p2, target_val_data)
            if use_latent_regularization:
                a, b = _reg_terms(p1, p2, y1_val, y2_val)
                reg_val = current_reg_weight * (0.9 * a + 0.1 * b)
                total_val_loss = LC_alpha * recon_loss_val + reg_val
                val_latent_reg_loss += reg_val
            else:
                total_val_loss = recon_loss_val
            val_loss += total_val_loss
            val_recon_loss += recon_loss_val
            val_batches += 1
        avg_val_loss = val_loss / val_batches if val_batches > 0 else 0.0
        avg_val_recon_loss = val_recon_loss / val_batches if val_batches > 0 else 0.0
        avg_val_latent_reg_loss = val_latent_reg_loss / current_reg_weight / val_batches if val_batches > 0 else 0.0
        current_lr = sched(epoch + 1)              # after this epoch's lr_scheduler.step()
        if avg_val_loss < best_val_loss:
            best_val_loss = avg_val_loss
            best_model_state = latent_conditioner.state_dict().copy()
            patience_counter = 0
        else:
            patience_counter += 1
        epoch_duration = time.time() - epoch_start_time
        avg_gradient_norm = epoch_gradient_sum / gradient_count if gradient_count > 0 else 0.0
        reg_weight_info = f", RegW: {current_reg_weight:.4f}" if use_latent_regularization else ""
        print("[%d/%d]\tTrain: %.4E (recon:%.4E, reg:%.4E), Val: %.4E (recon:%.4E, reg:%.4E), LR: %.2E (%s)%s, AvgGrad: %.4E, Best: %.4E, ETA: %.2f h" %
              (epoch, latent_conditioner_epoch, avg_train_loss, avg_train_recon_loss, avg_train_latent_reg_loss,
               avg_val_loss, avg_val_recon_loss, avg_val_latent_reg_loss, current_lr, "CosineAnnealing", reg_weight_info,
               avg_gradient_norm, best_val_loss, (latent_conditioner_epoch - epoch) * epoch_duration / 3600))
    os.makedirs("checkpoints", exist_ok=True)
    os.makedirs("model_save", exist_ok=True)
    torch.save(latent_conditioner.state_dict(), "checkpoints/latent_conditioner_e2e_improved.pth")
    with open("model_save/LatentConditioner", "wb") as f:
        pickle.dump(latent_conditioner, f)
    return avg_val_loss
