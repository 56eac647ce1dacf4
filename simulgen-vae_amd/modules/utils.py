"""Config parsing, distributed set-up and the resident dataset of the training path
(reference modules/utils.py:38-76,209-238,255-351,673-683)."""
from __future__ import annotations

import os
import shutil

import numpy as np
import torch
import torch.distributed as dist


def parse_condition_file(filepath):
    """`key value` lines; `#` starts a comment; lines starting with % or ' are skipped; the last
    duplicate key wins (utils.py:270-282)."""
    params = {}
    with open(filepath, encoding="utf-8") as f:
        for line in f:
            line = line.split("#")[0].strip()
            if not line or line.startswith("%") or line.startswith("'"):
                continue
            parts = line.split()
            if len(parts) >= 2:
                params[parts[0]] = parts[1]
    return params


def parse_training_parameters(params):
    """Typed extraction with the reference's keys and defaults (utils.py:302-351).  Missing mandatory
    keys raise KeyError exactly like the reference's dict lookups."""
    c = {}
    c["num_param"] = int(params["Dim1"])
    c["num_time"] = int(params["Dim2"])
    c["num_time_to"] = int(params["Dim2_red"])
    c["num_node"] = int(params["Dim3"])
    c["num_node_start"] = int(params["Dim3_start"])
    c["num_node_end"] = int(params["Dim3_end"])
    c["num_var"] = int(params["num_var"])
    c["n_epochs"] = int(params["Training_epochs"])
    c["batch_size"] = int(params["Batch_size"])
    c["LR"] = float(params["LearningR"])
    c["latent_dim"] = int(params["Latent_dim"])
    c["latent_dim_end"] = int(params["Latent_dim_end"])
    c["loss_type"] = int(params["Loss_type"])
    c["stretch"] = int(params["Stretch"])
    c["alpha"] = int(params["alpha"])
    c["num_samples_f"] = int(params.get("num_aug_f", 0))
    c["num_samples_a"] = int(params.get("num_aug_a", 0))
    c["recon_iter"] = int(params.get("Recon_iter", 1))
    c["num_physical_param"] = int(params["num_param"])
    c["param_dir"] = params["param_dir"]
    c["latent_conditioner_epoch"] = int(params["n_epoch"])
    c["latent_conditioner_lr"] = float(params["latent_conditioner_lr"])
    c["latent_conditioner_batch_size"] = int(params["latent_conditioner_batch"])
    c["latent_conditioner_data_type"] = params["input_type"]
    c["param_data_type"] = params["param_data_type"]
    c["latent_conditioner_weight_decay"] = float(params.get("latent_conditioner_weight_decay", 1e-4))
    c["latent_conditioner_dropout_rate"] = float(params.get("latent_conditioner_dropout_rate", 0.3))
    c["use_spatial_attention"] = int(params.get("use_spatial_attention", 1))
    c["use_e2e_training"] = int(params.get("use_e2e_training", 0))
    c["use_improved_e2e"] = int(params.get("use_improved_e2e", 0))
    c["e2e_loss_function"] = params.get("e2e_loss_function", "MSE")
    c["e2e_vae_model_path"] = params.get("e2e_vae_model_path", "model_save/SimulGen-VAE")
    c["use_latent_regularization"] = int(params.get("use_latent_regularization", 0))
    c["LC_alpha"] = float(params.get("LC_alpha", 1.0))
    c["latent_reg_weight"] = float(params.get("latent_reg_weight", 0.001))
    return c


LOSS_NAMES = {1: "MSE", 2: "MAE", 3: "smoothL1", 4: "Huber"}   # SimulGen-VAE.py:208-215


def read_preset(path, preset="1"):
    """preset.txt positional lines (SimulGen-VAE.py:197-204): header, data_No, init_beta_divisor,
    num_filter_enc, latent_conditioner_filter."""
    with open(path) as f:
        lines = [ln.strip() for ln in f.readlines()]
    return dict(data_No=int(lines[1]), init_beta_divisor=int(lines[2]),
                num_filter_enc=[int(v) for v in lines[3].split()],
                latent_conditioner_filter=[int(v) for v in lines[4].split()])


def setup_distributed_training(args):
    """utils.py:209-238: under torchrun, bind the local GPU and join the process group.  Backend is
    `nccl` (= RCCL on ROCm) on GPUs; `gloo` when no GPU is visible (CPU tests of the host logic)."""
    if not getattr(args, "use_ddp", False):
        return False
    try:
        local_rank = int(os.environ.get("LOCAL_RANK", -1))
        if local_rank == -1:
            print("For DDP training, please use: torchrun --nproc_per_node=NUM_GPUS SimulGen-VAE.py --use_ddp [other args]")
            return False
        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl")
        else:
            dist.init_process_group(backend="gloo")
        print(f"Initialized DDP process group. Rank {dist.get_rank()} of {dist.get_world_size()}")
        return True
    except Exception as e:  # same fallback as the reference
        print(f"Failed to initialize DDP: {e}")
        print("Falling back to single GPU training")
        return False


def initialize_folder(folder_name):
    os.makedirs(folder_name, exist_ok=True)
    for item in os.listdir(folder_name):
        p = os.path.join(folder_name, item)
        if os.path.isdir(p):
            shutil.rmtree(p)
        else:
            os.remove(p)


class Dataset(torch.utils.data.Dataset):
    """utils.Dataset (utils.py:38-76): with load_all the whole [P, num_node, num_time] array lives on the
    device; indexing returns one sample."""

    def __init__(self, x_data, load_all):
        self.load_all = bool(load_all)
        if hasattr(x_data, "buf") and hasattr(x_data, "num_node"):      # data_preprocess.DeviceDataset: already in HBM
            self.x_data = x_data
        elif self.load_all and torch.cuda.is_available():
            self.x_data = torch.as_tensor(np.asarray(x_data), dtype=torch.float32).cuda()
        else:
            self.x_data = np.ascontiguousarray(x_data)

    def __getitem__(self, index):
        if torch.is_tensor(self.x_data) or hasattr(self.x_data, "buf"):
            return self.x_data[index]
        return torch.from_numpy(self.x_data[index].copy()).float()

    def __len__(self):
        return len(self.x_data)


def _to_dev(a):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32).cuda().contiguous()


class LatentConditionerDataset(torch.utils.data.Dataset):
    """utils.LatentConditionerDataset (utils.py:120-173): (condition input, main latent, hierarchical latents) triples
    preloaded to the device; NaNs are replaced by zeros with the reference's warnings."""

    def __init__(self, input_data, output_data1, output_data2, load_all=True):
        arrs = []
        for nm, a in (("input_data", input_data), ("output_data1", output_data1), ("output_data2", output_data2)):
            a = np.asarray(a)
            if np.isnan(a).any():
                print(f"Warning: NaN values detected in {nm}, replacing with zeros")
                a = np.nan_to_num(a, nan=0.0)
            arrs.append(a)
        self.input_data, self.output_data1, self.output_data2 = (_to_dev(a) for a in arrs)
        self.on_gpu = True

    def __len__(self):
        return len(self.input_data)

    def __getitem__(self, idx):
        return self.input_data[idx], self.output_data1[idx], self.output_data2[idx]


class E2ELatentConditionerDataset(torch.utils.data.Dataset):
    """utils.E2ELatentConditionerDataset (utils.py:602-671): (condition, main latent, hierarchical latents,
    reconstruction target [num_node, num_time]) per sample.  With load_all everything is resident in HBM; otherwise the
    reconstruction targets (the big array) stay in host memory and are uploaded per batch."""

    def __init__(self, condition_data, latent_main_data, latent_hier_data, target_reconstruction_data, load_all=False):
        self.length = len(condition_data)
        self.load_all = bool(load_all)
        self.condition_data, self.latent_main_data, self.latent_hier_data = (_to_dev(a) for a in (condition_data, latent_main_data, latent_hier_data))
        if hasattr(target_reconstruction_data, "buf") and hasattr(target_reconstruction_data, "num_node"):
            self.target_reconstruction_data = target_reconstruction_data         # data_preprocess.DeviceDataset
        elif self.load_all:
            self.target_reconstruction_data = _to_dev(target_reconstruction_data)
            print(f"E2E Dataset loaded to GPU: {self.length} samples")
        else:
            self.target_reconstruction_data = torch.from_numpy(np.ascontiguousarray(target_reconstruction_data, dtype=np.float32))
            print(f"E2E Dataset loaded to CPU with pin_memory: {self.length} samples")

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return self.condition_data[idx], self.latent_main_data[idx], self.latent_hier_data[idx], self.target_reconstruction_data[idx]


def shard_indices(indices, rank, world, batch_size):
    """Data-parallel sharding: rank r takes indices r::world of the (already shuffled) list,
    trimmed so that every rank sees the same number of full-or-partial batches."""
    per = len(indices) // world
    return [indices[i * world + rank] for i in range(per)]


# ------------------------------------------------------------------------------------------------
# Post-training evaluation / latent export (reference modules/utils.py:428-596, called from
# SimulGen-VAE.py:311-344).  Same arguments, prints, return tuples and quirks: one row per *batch* (only the
# first sample of every batch is stored, exactly as the reference indexes `[0]`), `loss_total` is the sum of
# the per-batch MSEs, reconstruction PNGs for the first ten batches when matplotlib is importable.
# ------------------------------------------------------------------------------------------------
def reparameterize(mu, std, eps=None):
    """modules/decoder.py:218-223: z = mu + eps * clamp(std, 1e-8, 10); eps ~ N(0,1) unless injected."""
    std = torch.clamp(std, min=1e-8, max=10.0)
    if eps is None:
        eps = torch.randn_like(std)
    return mu + eps * std


def _mse(a, b):
    return torch.mean((a.float() - b.float()) ** 2)


def _dataset_shape(dataloader):
    ds = dataloader.dataset
    if hasattr(ds, "dataset"):
        ds = ds.dataset
    for attr in ("x_data", "data"):
        if hasattr(ds, attr):
            return tuple(getattr(ds, attr).shape)
    first = next(iter(dataloader))
    return tuple(first.shape)


def evaluate_vae_reconstruction(VAE, dataloader, device, num_param, num_filter_enc, latent_dim, latent_dim_end,
                                recon_iter=1, dataset_name="Dataset", save_images=True, eps_fn=None):
    """Encoder -> reparameterise -> `mode='fix'` decoder -> per-batch MSE, keeping the best of `recon_iter` draws.
    Returns (latent_vectors [num_param, latent_dim_end], hierarchical_latent_vectors [num_param, L-1, latent_dim],
    reconstruction_loss [num_param], reconstructed [num_param, num_node, num_time], loss_total).
    `eps_fn(j, i, like)` (not in the reference) injects the reparameterisation noise for parity tests."""
    save_dir = None
    if save_images:
        os.makedirs("checkpoints", exist_ok=True)
        save_dir = "checkpoints/" + dataset_name.replace(" ", "_").replace("(", "").replace(")", "").lower()
        os.makedirs(save_dir, exist_ok=True)
    latent_vectors = np.zeros([num_param, latent_dim_end])
    hierarchical_latent_vectors = np.zeros([num_param, len(num_filter_enc) - 1, latent_dim])
    reconstruction_loss = np.zeros([num_param])
    data_shape = _dataset_shape(dataloader)
    reconstructed = np.empty([num_param, data_shape[1], data_shape[2]])
    loss_total = 0
    print(f"Evaluating {dataset_name}...")
    j = -1
    for j, image in enumerate(dataloader):
        best = 100.0
        x = VAE._prep(image) if hasattr(VAE, "_prep") else image.to(device)
        mu, log_var, xs = VAE.encoder(x)
        loss = None
        for i in range(recon_iter):
            std = torch.exp(0.5 * log_var)
            latent_vector = reparameterize(mu, std, None if eps_fn is None else eps_fn(j, i, std))
            gen_x, _ = VAE.decoder(latent_vector, xs, mode="fix")
            loss = _mse(gen_x, x)
            if float(loss) < best:
                best = float(loss)
                latent_vectors[j, :] = latent_vector[0, :].detach().cpu().numpy()
                for k in range(len(xs)):
                    hierarchical_latent_vectors[j, k, :] = xs[k].detach().cpu().numpy()[0]
                reconstruction_loss[j] = best
                reconstructed[j, :, :] = gen_x[0].detach().float().cpu().numpy()
        print(f"Parameter {j + 1} finished - MSE: {float(loss):.4E}")
        loss_total = loss_total + float(loss)
        if save_images and j < 10:
            try:
                import matplotlib
                matplotlib.use("Agg")
                import matplotlib.pyplot as plt
                original = x[0].detach().float().cpu().numpy()
                recon = reconstructed[j]
                plt.figure(figsize=(12, 6))
                nplot = min(3, original.shape[0])
                for ch in range(nplot):
                    plt.subplot(nplot, 1, ch + 1)
                    plt.plot(original[ch], label="Original", alpha=0.7)
                    plt.plot(recon[ch], label="Reconstructed", alpha=0.7, linestyle="--")
                    plt.title(f"Channel {ch + 1} - Sample {j + 1} - MSE: {float(loss):.4E}")
                    plt.legend()
                    plt.grid(True, alpha=0.3)
                plt.tight_layout()
                plt.savefig(f"{save_dir}/reconstruction_sample_{j + 1:03d}.png", dpi=300, bbox_inches="tight")
                plt.close()
            except Exception as e:  # same degradation as the reference
                print(f"Warning: Could not save reconstruction image for sample {j + 1}: {e}")
    print("")
    average_loss = loss_total / (j + 1) if j >= 0 else 0
    print(f"Total {dataset_name} MSE loss: {average_loss:.3e}")
    if save_images:
        print(f"Saved {min(10, j + 1) if j >= 0 else 0} reconstruction images to: {save_dir}/")
    print("--------------------------------")
    print("")
    return latent_vectors, hierarchical_latent_vectors, reconstruction_loss, reconstructed, loss_total


def evaluate_vae_simple(VAE, dataloader, device, dataset_name="Dataset", eps_fn=None):
    """utils.py:563-596: sum of per-batch reconstruction MSEs, nothing stored."""
    loss_total = 0
    print(f"Evaluating {dataset_name}...")
    j = -1
    for j, image in enumerate(dataloader):
        x = VAE._prep(image) if hasattr(VAE, "_prep") else image.to(device)
        mu, log_var, xs = VAE.encoder(x)
        std = torch.exp(0.5 * log_var)
        z = reparameterize(mu, std, None if eps_fn is None else eps_fn(j, 0, std))
        gen_x, _ = VAE.decoder(z, xs, mode="fix")
        loss = _mse(gen_x, x)
        print(f"Parameter {j + 1} finished - MSE: {float(loss):.4E}")
        loss_total = loss_total + float(loss)
    print("")
    average_loss = loss_total / (j + 1) if j >= 0 else 0
    print(f"Total {dataset_name} MSE loss: {average_loss:.3e}")
    print("--------------------------------")
    print("")
    return loss_total


def export_latents(VAE, x_all, num_filter_enc, latent_dim, latent_dim_end, recon_iter=1, out_dir="model_save",
                   loss_file="./SimulGen-VAE_L2_loss.txt", save_images=False, eps_fn=None):
    """The "whole dataset" leg of SimulGen-VAE.py:325-344: batch-1 pass over every sample, then
    `model_save/latent_vectors.npy`, `model_save/xs.npy` and `SimulGen-VAE_L2_loss.txt` in the reference's formats
    (what `--lc_only=1` later loads, SimulGen-VAE.py:348-350)."""
    loader = torch.utils.data.DataLoader(Dataset(x_all, False), batch_size=1, shuffle=False, num_workers=0)
    lat, hier, rl, _, _ = evaluate_vae_reconstruction(VAE, loader, "cuda", len(x_all), num_filter_enc, latent_dim,
                                                      latent_dim_end, recon_iter, "Whole Dataset", save_images, eps_fn)
    os.makedirs(out_dir, exist_ok=True)
    np.save(os.path.join(out_dir, "latent_vectors"), lat)
    np.save(os.path.join(out_dir, "xs"), hier)
    np.savetxt(loss_file, rl, fmt="%e")
    return lat, hier, rl
