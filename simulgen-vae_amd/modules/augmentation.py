"""On-the-fly augmentation and loaders of the training path (reference modules/augmentation.py:9-242),
with the per-sample torch ops + default collate replaced by ONE fused augment+collate kernel on the
HBM-resident dataset (sgv_augment_collate).  Same defaults: noise p=.5 sigma=.05, scale p=.5 in [.9,1.1],
mixup p=.5 alpha=.2 (lambda clipped to [.1,.9]), shift/cutout off; the `augmentation_config` argument is
accepted and ignored exactly as in the reference (augmentation.py:14,26-38)."""
from __future__ import annotations

import random

import numpy as np
import torch

from .utils import Dataset, shard_indices


class AugmentedDataset(Dataset):
    def __init__(self, x_data, load_all, augmentation_config=None):
        super().__init__(x_data, load_all)
        self.augmentation_config = {
            "noise_prob": 0.5, "noise_level": 0.05, "scaling_prob": 0.5, "scaling_range": (0.9, 1.1),
            "shift_prob": 0.0, "shift_max": 0.0, "mixup_prob": 0.5, "mixup_alpha": 0.2,
            "cutout_prob": 0.0, "cutout_max": 0.0, "enabled": True,
        }
        self.training = True

    def plan(self, index, key=None):
        """The random draws of AugmentedDataset._apply_augmentations (augmentation.py:58-84) in the
        reference's order: python `random` for decisions/scale, numpy for the mixup lambda.  Returns
        (noise_seed, scale, mix_index, lam); noise itself is drawn on the device (Philox).
        key = None: the process-global streams, as the reference (single process).  key = an integer: generators of their own
        seeded with it -- the data-parallel loaders pass (job seed, epoch, position in the global shuffled list), so a sample's
        augmentation does not depend on how many ranks the list is dealt to."""
        cfg = self.augmentation_config
        if not cfg["enabled"] or not self.training:
            return 0, 1.0, -1, 1.0
        rnd = random if key is None else random.Random(key)
        seed = (rnd.getrandbits(62) | 1) if rnd.random() < cfg["noise_prob"] else 0
        scale = 1.0
        if rnd.random() < cfg["scaling_prob"]:
            lo, hi = cfg["scaling_range"]
            scale = lo + rnd.random() * (hi - lo)
        rnd.random()  # shift draw (probability 0)
        mix, lam = -1, 1.0
        if rnd.random() < cfg["mixup_prob"] and len(self) > 1:
            mix = rnd.randint(0, len(self) - 1)
            while mix == index:
                mix = rnd.randint(0, len(self) - 1)
            beta = np.random.beta if key is None else np.random.RandomState(key & 0xFFFFFFFF).beta
            lam = max(0.1, min(float(beta(cfg["mixup_alpha"], cfg["mixup_alpha"])), 0.9))
        rnd.random()  # cutout draw (probability 0)
        return seed, scale, mix, lam

    def __getitem__(self, index):
        """Per-sample path kept for API compatibility (host/torch ops, not used by modules.train.train)."""
        sample = super().__getitem__(index)
        seed, scale, mix, lam = self.plan(index)
        if seed:
            g = torch.Generator(device=sample.device).manual_seed(seed & 0x7FFFFFFF)
            sample = sample + torch.randn(sample.shape, generator=g, device=sample.device) * 0.05
        if scale != 1.0:
            sample = sample * scale
        if mix >= 0:
            sample = lam * sample + (1 - lam) * super().__getitem__(mix)
        return sample

    def set_training(self, training=True):
        self.training = training

    def set_augmentation_enabled(self, enabled=True):
        self.augmentation_config["enabled"] = enabled


class ResidentLoader:
    """DataLoader stand-in: iterating yields [B, num_node, num_time] tensors like the reference's loader;
    modules.train.train instead asks for `batch_plans()` and feeds them to the fused device kernel."""

    def __init__(self, dataset: AugmentedDataset, indices, batch_size, shuffle, augment, rank=0, world=1, shuffle_seed=None):
        self.dataset, self.indices = dataset, list(int(i) for i in indices)
        self.batch_size, self.shuffle, self.augment = int(batch_size), shuffle, augment
        self.rank, self.world = rank, world
        # data-parallel runs: the per-epoch permutation comes from a generator of its own, seeded with a value every rank
        # shares (drawn on rank 0, broadcast by create_augmented_dataloaders) -- never from the process-global `random`,
        # whose state differs between ranks (urandom seeding under torchrun, and the augmentation draws consume a
        # data-dependent number of values)
        self.shuffle_seed = shuffle_seed
        self.epoch = 0
        self._resident = None   # (engine id, device buffer)

    def __len__(self):
        n = len(self.indices) // self.world if self.world > 1 else len(self.indices)
        return (n + self.batch_size - 1) // self.batch_size

    def _epoch_indices(self):
        idx = list(self.indices)
        if self.shuffle:
            if self.shuffle_seed is not None:
                random.Random(self.shuffle_seed + self.epoch).shuffle(idx)     # identical on every rank
            else:
                random.shuffle(idx)                                             # single process: the reference's global stream
        self.epoch += 1
        if self.world > 1:
            idx = shard_indices(idx, self.rank, self.world, self.batch_size)
        return idx

    def batch_plans(self):
        idx = self._epoch_indices()
        epoch = self.epoch - 1
        for i in range(0, len(idx), self.batch_size):
            b = idx[i:i + self.batch_size]
            was = self.dataset.augmentation_config["enabled"]
            self.dataset.augmentation_config["enabled"] = was and self.augment
            if self.shuffle_seed is not None:
                # keyed by the position in the GLOBAL shuffled list (local position k of rank r is k * world + r): world-size-invariant
                keys = [(self.shuffle_seed * 1000003 + epoch) * 2147483647 + (i + k) * self.world + self.rank for k in range(len(b))]
                plans = [self.dataset.plan(j, key) for j, key in zip(b, keys)]
            else:
                plans = [self.dataset.plan(j) for j in b]
            self.dataset.augmentation_config["enabled"] = was
            yield b, [p[0] for p in plans], [p[1] for p in plans], [p[2] for p in plans], [p[3] for p in plans]

    def resident(self, engine):
        """Dataset converted once to the engine's layout/dtype and kept in HBM (utils.Dataset load_all)."""
        x = self.dataset.x_data
        if hasattr(x, "buf") and hasattr(x, "num_node"):     # data_preprocess.DeviceDataset: written in this layout already
            from ..engine import DTYPES
            if DTYPES[x.dtype] != DTYPES[engine.compute_dtype] or x.buf.numel() != len(x) * engine.sample_bytes():
                raise ValueError(f"device dataset is {x.dtype} {x.shape}, engine computes in {engine.compute_dtype}")
            return x.buf
        if self._resident is None or self._resident[0] is not engine:
            P = len(self.dataset)
            buf = torch.empty(P * engine.sample_bytes(), dtype=torch.uint8, device="cuda")
            x = self.dataset.x_data
            step = 8
            for p0 in range(0, P, step):
                c = min(step, P - p0)
                src = x[p0:p0 + c] if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x[p0:p0 + c]))
                src = src.to(device="cuda", dtype=torch.float32).contiguous()
                engine.dataset_convert(src, buf[p0 * engine.sample_bytes():], c)
            torch.cuda.synchronize()
            self._resident = (engine, buf)
        return self._resident[1]

    def __iter__(self):
        for b, seeds, scale, mix, lam in self.batch_plans():
            out = []
            for j, s, sc, m, l in zip(b, seeds, scale, mix, lam):
                x = Dataset.__getitem__(self.dataset, j)
                if s:
                    g = torch.Generator(device=x.device).manual_seed(s & 0x7FFFFFFF)
                    x = x + torch.randn(x.shape, generator=g, device=x.device) * 0.05
                x = x * sc
                if m >= 0:
                    x = l * x + (1 - l) * Dataset.__getitem__(self.dataset, m)
                out.append(x)
            yield torch.stack(out)


def create_augmented_dataloaders(x_data, batch_size, load_all=False, augmentation_config=None, val_split=0.2,
                                 num_workers=None):
    """augmentation.py:151-242: torch.randperm 80/20 split, shuffled augmented train loader, plain val loader.
    Under torch.distributed the train indices are sharded r::world per epoch."""
    import torch.distributed as dist
    dataset_size = len(x_data)
    val_size = int(dataset_size * val_split)
    train_size = dataset_size - val_size
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    indices = torch.randperm(dataset_size)
    shuffle_seed = None
    if world > 1:
        # one split and one shuffle seed for the whole job: rank 0's draws, broadcast (the reference seeds nothing, so every
        # rank's own draws would differ: validation samples of one rank would be trained on by another)
        box = [indices.tolist(), random.getrandbits(48)] if rank == 0 else [None, None]
        dist.broadcast_object_list(box, src=0)
        indices, shuffle_seed = torch.tensor(box[0], dtype=torch.long), int(box[1])
    train_indices, val_indices = indices[:train_size].tolist(), indices[train_size:].tolist()
    full = AugmentedDataset(x_data, load_all, augmentation_config)
    train_loader = ResidentLoader(full, train_indices, batch_size, shuffle=True, augment=True, rank=rank, world=world,
                                  shuffle_seed=shuffle_seed)
    val_loader = ResidentLoader(full, val_indices, batch_size, shuffle=False, augment=False)
    return train_loader, val_loader
