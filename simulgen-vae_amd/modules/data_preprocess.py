"""Input pipeline of the training path on the GPU: the reference's `reduce_dataset` and
`data_scaler` (modules/data_preprocess.py:13-41,65-165, called from SimulGen-VAE.py:267-283).

`data_scaler` keeps the reference's signature, prints, sampling (np.random.seed(42) + choice without
replacement over the P*T rows), return tuple `(new_x_train [P,T,N], DATA_shape, scaler)` and `model_save/
scaler.pkl`, but fits the per-node MinMaxScaler(-0.7, 0.7) and transforms on the MI355X (`sgv_minmax_fit`,
`sgv_minmax_coeffs`, `sgv_scale_convert`).  With `device_dataset=True` the scaled data never returns to the
host: it is written once, in the engine's resident layout and compute dtype, and `create_augmented_dataloaders`
/ `train` use it as is (the reference's transpose to [P,N,T] and the engine's transpose back cancel)."""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np
import torch

from ..engine import DTYPES, SgvError, load_library

FEATURE_RANGE = (-0.7, 0.7)


def reduce_dataset(data_save, num_time_to, num_node_red, num_param, num_time, num_node_red_start, num_node_red_end):
    """data_preprocess.py:13-41: optional crop to [0:num_time_to] x [node_start:node_end] (fp64 zeros buffer as in
    the reference when a crop happens)."""
    start = time.time()
    num_node = data_save.shape[-1]
    if num_time_to == num_time and num_node_red == num_node:
        FOM_data = data_save
    else:
        num_time = num_time_to
        FOM_data = np.zeros((num_param, num_time, num_node_red))
        FOM_data[:, 0:num_time, :] = data_save[:, 0:num_time, num_node_red_start:num_node_red_end]
        num_node = num_node_red
    print()
    print()
    print(f"Dataset reduction completed in {time.time() - start:.2f}s")
    print(f"Dataset reduced to FOM data shape: {FOM_data.shape}")
    return num_time, FOM_data, num_node


class DeviceDataset:
    """Scaled dataset resident in HBM in the engine layout [P][T][N] (compute dtype).  Indexing returns the
    reference-shaped [num_node, num_time] fp32 sample, so it can stand in for the numpy array the reference
    passes around (`len`, `.shape`, `[i]`)."""

    def __init__(self, buf, P, num_node, num_time, dtype):
        self.buf, self.P, self.num_node, self.num_time, self.dtype = buf, int(P), int(num_node), int(num_time), dtype
        self.shape = (self.P, self.num_node, self.num_time)

    def __len__(self):
        return self.P

    def _typed(self):
        t = torch.bfloat16 if DTYPES[self.dtype] == 1 else torch.float32
        return self.buf.view(t).view(self.P, self.num_time, self.num_node)

    def __getitem__(self, i):
        return self._typed()[i].transpose(-1, -2).to(torch.float32)


def _sample_rows(total_samples):
    """data_preprocess.py:94-109."""
    max_samples = min(50000, total_samples // 10)
    if max_samples < 1000:
        max_samples = min(1000, total_samples)
    sample_stride = max(1, total_samples // max_samples)
    np.random.seed(42)
    if total_samples > max_samples:
        idx = np.random.choice(total_samples, max_samples, replace=False)
    else:
        idx = np.arange(total_samples)
    return idx, max_samples, sample_stride


class GpuMinMaxScaler:
    """Per-node MinMaxScaler with sklearn's attribute names (`data_min_`, `data_max_`, `data_range_`, `scale_`,
    `min_`, `feature_range`, `n_features_in_`, `n_samples_seen_`); `to_sklearn()` builds the real object."""

    def __init__(self, feature_range=FEATURE_RANGE):
        self.feature_range = tuple(feature_range)
        self._lib = load_library()
        self._d = None

    def _check(self, rc, what):
        if rc != 0:
            raise SgvError(f"{what}: {self._lib.sgv_last_error().decode()}")

    def fit_rows(self, rows, chunk_rows=2048):
        """rows: [S, N] host array (any float dtype) or CUDA fp32 tensor."""
        S, N = rows.shape
        mn = torch.empty(N, dtype=torch.float32, device="cuda")
        mx = torch.empty(N, dtype=torch.float32, device="cuda")
        for r0 in range(0, S, chunk_rows):
            part = rows[r0:r0 + chunk_rows]
            if not torch.is_tensor(part):
                part = torch.from_numpy(np.ascontiguousarray(part, dtype=np.float32))
            part = part.to(device="cuda", dtype=torch.float32).contiguous()
            self._check(self._lib.sgv_minmax_fit(C.c_void_p(part.data_ptr()), part.shape[0], N, C.c_void_p(mn.data_ptr()),
                                                 C.c_void_p(mx.data_ptr()), int(r0 > 0), None), "sgv_minmax_fit")
        sc = torch.empty(N, dtype=torch.float32, device="cuda")
        of = torch.empty(N, dtype=torch.float32, device="cuda")
        self._check(self._lib.sgv_minmax_coeffs(C.c_void_p(mn.data_ptr()), C.c_void_p(mx.data_ptr()), N,
                                                float(self.feature_range[0]), float(self.feature_range[1]),
                                                C.c_void_p(sc.data_ptr()), C.c_void_p(of.data_ptr()), None), "sgv_minmax_coeffs")
        torch.cuda.synchronize()
        self._d = (sc, of)
        self.data_min_, self.data_max_ = mn.cpu().numpy(), mx.cpu().numpy()
        self.data_range_ = self.data_max_ - self.data_min_
        self.scale_, self.min_ = sc.cpu().numpy(), of.cpu().numpy()
        self.n_features_in_, self.n_samples_seen_ = N, S
        return self

    def transform_to(self, src, dst, dtype):
        """src: CUDA fp32 [R, N] raw rows; dst: CUDA buffer for R*N values of `dtype` ("f32" | "bf16")."""
        R, N = src.shape
        self._check(self._lib.sgv_scale_convert(DTYPES[dtype], C.c_void_p(src.data_ptr()), C.c_void_p(self._d[0].data_ptr()),
                                                C.c_void_p(self._d[1].data_ptr()), C.c_void_p(dst.data_ptr()), R, N, None),
                    "sgv_scale_convert")

    def to_sklearn(self):
        from sklearn.preprocessing import MinMaxScaler
        s = MinMaxScaler(feature_range=self.feature_range)
        for k in ("data_min_", "data_max_", "data_range_", "scale_", "min_", "n_features_in_", "n_samples_seen_"):
            setattr(s, k, getattr(self, k))
        return s


def data_scaler(FOM_data_aug, FOM_data, num_time, num_node, directory, chunk_size=None, device_dataset=False,
                compute_dtype="bf16", params_per_chunk=8):
    start = time.time()
    if chunk_size is None:
        chunk_size = 10000
    print()
    print()
    print(f"Fitting scaler on dataset of shape: {FOM_data_aug.shape}")
    P = FOM_data_aug.shape[0]
    total_samples = P * FOM_data_aug.shape[1]
    sample_indices, max_samples, sample_stride = _sample_rows(total_samples)
    print(f"Sampling {max_samples} representative samples (every {sample_stride}th sample)")
    param_indices = sample_indices // num_time
    time_indices = sample_indices % num_time
    scaler = GpuMinMaxScaler(FEATURE_RANGE)
    # gather the sampled rows in slices so the host never holds more than ~1 GB of them
    rows_per = max(1, (1 << 28) // max(1, num_node))
    first = True
    mn = mx = None
    S = len(sample_indices)
    for r0 in range(0, S, rows_per):
        rows = FOM_data_aug[param_indices[r0:r0 + rows_per], time_indices[r0:r0 + rows_per], :]
        part = GpuMinMaxScaler(FEATURE_RANGE).fit_rows(rows)
        mn = part.data_min_ if first else np.fmin(mn, part.data_min_)
        mx = part.data_max_ if first else np.fmax(mx, part.data_max_)
        first = False
    scaler.fit_rows(np.stack([mn, mx]))                  # min/max of (min, max) rows == the global min/max
    scaler.n_samples_seen_ = S
    print("Transforming training data in chunks...")
    T = FOM_data_aug.shape[1]
    esz = 2 if DTYPES[compute_dtype] == 1 else 4
    dev = torch.empty(P * T * num_node * esz, dtype=torch.uint8, device="cuda") if device_dataset else None
    out = None if device_dataset else np.empty(FOM_data_aug.shape, dtype=np.float32)
    stage = torch.empty((params_per_chunk * T, num_node), dtype=torch.float32, device="cuda")
    n_chunks = (P + params_per_chunk - 1) // params_per_chunk
    for ci, p0 in enumerate(range(0, P, params_per_chunk)):
        c = min(params_per_chunk, P - p0)
        src = torch.from_numpy(np.ascontiguousarray(FOM_data_aug[p0:p0 + c], dtype=np.float32)).cuda().view(c * T, num_node)
        if device_dataset:
            scaler.transform_to(src, dev[p0 * T * num_node * esz:], compute_dtype)
        else:
            scaler.transform_to(src, stage, "f32")
            out[p0:p0 + c] = stage[:c * T].view(c, T, num_node).cpu().numpy()
        if ci % 5 == 0 or p0 + c == P:
            print(f"  Progress: {(p0 + c) / P * 100:.1f}% ({ci + 1}/{n_chunks} chunks)")
    torch.cuda.synchronize()
    DATA_shape = tuple(FOM_data_aug.shape[1:])
    os.makedirs("./model_save", exist_ok=True)
    try:
        from pickle import dump
        with open("./model_save/scaler.pkl", "wb") as f:
            dump(scaler.to_sklearn(), f)
        print("   Scaler saved to: ./model_save/scaler.pkl")
    except ImportError:
        print("   sklearn not importable: scaler.pkl not written")
    print(f"   Data scaling completed in {time.time() - start:.2f} seconds")
    if device_dataset:
        new_x_train = DeviceDataset(dev, P, num_node, T, compute_dtype)
        print(f"   Final data shape: {(P, T, num_node)}, resident in HBM as {compute_dtype} [P][T][N]")
    else:
        new_x_train = out
        print(f"   Final data shape: {new_x_train.shape}, dtype: {new_x_train.dtype}")
    return new_x_train, DATA_shape, scaler


def latent_conditioner_scaler(data, name):
    """data_preprocess.py:167-195: MinMaxScaler(-0.7, 0.7) over the rows of a [P, ...] array (3-D arrays are flattened
    per sample), pickled to `name`; returns (scaled_data, scaler).  The arrays here are the exported latents
    ([P, 32] and [P, 24] at preset 1): the host fit is the reference's own and costs microseconds, the scaler object
    is sklearn's so `latent_vectors_scaler.pkl` / `xs_scaler.pkl` stay loadable by the reference's loops."""
    from pickle import dump
    from sklearn.preprocessing import MinMaxScaler
    scaler = MinMaxScaler(feature_range=FEATURE_RANGE)
    original_shape = data.shape
    if original_shape[0] == 0:
        raise ValueError(f"Empty data array detected with shape {original_shape}. "
                         "Please check your data loading configuration. "
                         "If using 'input_type image', ensure PNG files exist in the specified directory.")
    data_reshaped = data.reshape(original_shape[0], -1) if len(original_shape) == 3 else data
    scaler.fit(data_reshaped)
    scaled_data = scaler.transform(data_reshaped)
    if len(original_shape) == 3:
        scaled_data = scaled_data.reshape(original_shape)
    with open(name, "wb") as f:
        dump(scaler, f)
    return scaled_data, scaler
