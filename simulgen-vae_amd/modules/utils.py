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
        if self.load_all and torch.cuda.is_available():
            self.x_data = torch.as_tensor(np.asarray(x_data), dtype=torch.float32).cuda()
        else:
            self.x_data = np.ascontiguousarray(x_data)

    def __getitem__(self, index):
        if torch.is_tensor(self.x_data):
            return self.x_data[index]
        return torch.from_numpy(self.x_data[index].copy()).float()

    def __len__(self):
        return len(self.x_data)


def shard_indices(indices, rank, world, batch_size):
    """Data-parallel sharding: rank r takes indices r::world of the (already shuffled) list,
    trimmed so that every rank sees the same number of full-or-partial batches."""
    per = len(indices) // world
    return [indices[i * world + rank] for i in range(per)]
