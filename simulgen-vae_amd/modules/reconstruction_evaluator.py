"""`ReconstructionEvaluator` with the surface of the reference's modules/reconstruction_evaluator.py:18-274 (SURVEY 8(f)
N4): per sample, the conditioner's predicted latents and the true latents are descaled (MinMaxScaler.inverse_transform),
decoded by the VAE decoder in mode='fix' (`sgv_decode`) and compared with the original field -- dual-view PNGs under
`checkpoints/reconstruction_dual_view_<i>.png` and, with debug_mode >= 1, the reference's statistics printout.
Conditioner forward, descaling and decoding run on the MI355X; matplotlib (Agg) draws on the host from the three arrays
the reference plots.  `plt.show()` has no counterpart (no display)."""
from __future__ import annotations

import os

import numpy as np
import torch

from .latent_conditioner import _dev
from .latent_conditioner_e2e import descale_latent_predictions


class ReconstructionEvaluator:
    def __init__(self, VAE, device, num_time, debug_mode=0):
        self.VAE = VAE
        self.device = device
        self.num_time = num_time
        self.debug_mode = debug_mode

    # reconstruction_evaluator.py:35-104: batch-size-1, unshuffled walk over the conditioner dataset and the original data
    def evaluate_reconstruction_comparison(self, latent_conditioner, latent_conditioner_dataset, original_data,
                                           latent_vectors_scaler, xs_scaler):
        if self.debug_mode >= 1:
            print(f"Evaluating {len(latent_conditioner_dataset)} samples...")
        latent_conditioner.eval()
        n = min(len(latent_conditioner_dataset), len(original_data))
        for i in range(n):
            x_lc, y1_true, y2_true = latent_conditioner_dataset[i]
            x_orig = original_data[i]
            self._evaluate_one(i, latent_conditioner, _dev(x_lc)[None], _dev(y1_true)[None], _dev(y2_true)[None],
                               _dev(x_orig)[None], latent_vectors_scaler, xs_scaler)

    # reconstruction_evaluator.py:106-160: the same comparison over an E2E dataloader (x, y1, y2, x_orig); as in the
    # reference only the first sample of every batch is reconstructed and plotted
    def evaluate_reconstruction_comparison_e2e(self, latent_conditioner, dataloader_test, original_data,
                                               latent_vectors_scaler, xs_scaler):
        if self.debug_mode >= 1:
            print("Evaluating E2E reconstruction with DataLoader...")
            print(f"DataLoader type: {type(dataloader_test)}")
        latent_conditioner.eval()
        for i, (x_lc, y1_true, y2_true, x_orig) in enumerate(dataloader_test):
            self._evaluate_one(i, latent_conditioner, _dev(x_lc), _dev(y1_true), _dev(y2_true), _dev(x_orig),
                               latent_vectors_scaler, xs_scaler)

    def _evaluate_one(self, i, latent_conditioner, x_lc, y1_true, y2_true, x_orig, latent_vectors_scaler, xs_scaler):
        y_pred1, y_pred2 = latent_conditioner(x_lc)
        predicted = self._reconstruct_from_latents(y_pred1, y_pred2, latent_vectors_scaler, xs_scaler)
        true_recon = self._reconstruct_from_latents(y1_true, y2_true, latent_vectors_scaler, xs_scaler)
        original = x_orig.detach().cpu().numpy()
        self._plot_reconstruction_comparison(i, original, predicted, true_recon, save_plots=True)
        if self.debug_mode >= 1:
            self._print_reconstruction_stats(i, original, predicted, true_recon)

    def _reconstruct_from_latents(self, y_pred, y2_pred, latent_scaler, xs_scaler):
        """-> numpy [1, num_time, num_node] (the decoder output with axes 1 and 2 swapped, as the reference returns it).
        The reference flattens y2 to ONE row (`reshape([1, -1])`), i.e. it reconstructs the first sample of what it is
        given; so does this."""
        y1 = _dev(y_pred)[:1]
        y2 = _dev(y2_pred)
        d = y2.shape[-1]
        y2 = y2.reshape(y2.shape[0], -1)[:1] if y2.dim() > 1 else y2.reshape(1, -1)
        lat, xs = descale_latent_predictions(y1.contiguous(), y2.contiguous(), latent_scaler, xs_scaler)
        xs_list = [xs[:, k * d:(k + 1) * d].contiguous() for k in range(xs.shape[1] // d)]
        target_output, _ = self.VAE.decoder(lat, xs_list, mode="fix")
        return target_output.detach().cpu().numpy().swapaxes(1, 2)

    def _plot_reconstruction_comparison(self, sample_idx, original, predicted, true_recon, save_plots=False):
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, axes = plt.subplots(2, 2, figsize=(16, 12))
        fig.suptitle(f"Sample {sample_idx} - Dual-View Reconstruction Comparison", fontsize=16)
        num_nodes = original.shape[1]
        time_indices = [int(self.num_time * 0.25), int(self.num_time * 0.5), int(self.num_time * 0.75)]
        node_indices = [int(num_nodes * 0.25), int(num_nodes * 0.5), int(num_nodes * 0.75)]
        colors = ["blue", "green", "red"]
        rng_ = lambda a: f"[{a.min():.1f}, {a.max():.1f}]"

        time_idx = int(self.num_time / 2)
        o, p, t = original[0, :, time_idx] * 1e6, predicted[0, time_idx, :] * 1e6, true_recon[0, time_idx, :] * 1e6
        ax = axes[0, 0]
        ax.set_title(f"Nodal View - Spatial Distribution (t={time_idx})")
        ax.plot(o, ".", label=f"Original {rng_(o)}", alpha=0.8, markersize=1)
        ax.plot(p, ".", label=f"VAE+LC {rng_(p)}", alpha=0.8, markersize=1)
        ax.plot(t, ".", label=f"VAE-only {rng_(t)}", alpha=0.8, markersize=1)
        ax.set_xlabel("Node Index"); ax.set_ylabel("Value (x1e6)"); ax.legend(); ax.grid(True, alpha=0.3)

        ax = axes[0, 1]
        ax.set_title("Nodal View - Multiple Time Snapshots")
        for c, t_idx in zip(colors, time_indices):
            ax.plot(original[0, :, t_idx] * 1e6, "--", color=c, alpha=0.7, linewidth=1, label=f"Original t={t_idx}")
            ax.plot(predicted[0, t_idx, :] * 1e6, "-", color=c, alpha=0.8, linewidth=1, label=f"VAE+LC t={t_idx}")
        ax.set_xlabel("Node Index"); ax.set_ylabel("Value (x1e6)")
        ax.legend(bbox_to_anchor=(1.05, 1), loc="upper left"); ax.grid(True, alpha=0.3)

        node_idx = int(num_nodes / 2)
        o, p, t = original[0, node_idx, :] * 1e6, predicted[0, :, node_idx] * 1e6, true_recon[0, :, node_idx] * 1e6
        ax = axes[1, 0]
        ax.set_title(f"Temporal View - Time Evolution (node={node_idx})")
        ax.plot(o, "-", label=f"Original {rng_(o)}", alpha=0.8)
        ax.plot(p, "-", label=f"VAE+LC {rng_(p)}", alpha=0.8)
        ax.plot(t, "-", label=f"VAE-only {rng_(t)}", alpha=0.8)
        ax.set_xlabel("Time Index"); ax.set_ylabel("Value (x1e6)"); ax.legend(); ax.grid(True, alpha=0.3)

        ax = axes[1, 1]
        ax.set_title("Temporal View - Multiple Node Traces")
        for c, n_idx in zip(colors, node_indices):
            ax.plot(original[0, n_idx, :] * 1e6, "--", color=c, alpha=0.7, linewidth=1, label=f"Original n={n_idx}")
            ax.plot(predicted[0, :, n_idx] * 1e6, "-", color=c, alpha=0.8, linewidth=1, label=f"VAE+LC n={n_idx}")
        ax.set_xlabel("Time Index"); ax.set_ylabel("Value (x1e6)")
        ax.legend(bbox_to_anchor=(1.05, 1), loc="upper left"); ax.grid(True, alpha=0.3)

        plt.tight_layout()
        if save_plots:
            os.makedirs("checkpoints", exist_ok=True)
            plt.savefig(f"checkpoints/reconstruction_dual_view_{sample_idx}.png", dpi=300, bbox_inches="tight")
        plt.close(fig)

    def _print_reconstruction_stats(self, sample_idx, original, predicted, true_recon):
        time_idx = int(self.num_time / 2)
        original_slice = original[0, :, time_idx]
        predicted_slice = predicted[0, time_idx, :]
        true_recon_slice = true_recon[0, time_idx, :]
        pred_error = np.mean((original_slice - predicted_slice) ** 2)
        true_error = np.mean((original_slice - true_recon_slice) ** 2)
        print(f"Sample {sample_idx} Reconstruction Stats:")
        print(f"  Original range: [{original_slice.min():.3e}, {original_slice.max():.3e}]")
        print(f"  VAE+LC MSE: {pred_error:.3e}")
        print(f"  VAE-only MSE: {true_error:.3e}")
        print(f"  VAE-only should be ~0 (got {true_error:.1e})")
