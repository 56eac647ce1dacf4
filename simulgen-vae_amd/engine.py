"""ctypes binding of libsgvae.so (include/sgvae.h).

PyTorch is used only as plumbing here: device memory for the tensors handed across the C ABI
and the HIP stream the engine enqueues on.  There is no fallback: if the library or a GPU is
missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import numpy as np

from .spec import LOSS_IDS, VAEConfig, param_spec

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGV_LIB") or os.path.join(_HERE, "csrc", "libsgvae.so")
MAX_LEVELS = 8
MAX_SCALARS = 12
DTYPES = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}

# every symbol include/sgvae.h declares (tests/test_abi.py checks the library exports them all)
ABI_SYMBOLS = [
    "sgv_last_error", "sgv_create", "sgv_destroy", "sgv_param_count", "sgv_param_info", "sgv_load_state",
    "sgv_export_state", "sgv_export_grad", "sgv_export_adam", "sgv_prepare", "sgv_set_input", "sgv_set_eps",
    "sgv_seed", "sgv_set_shard", "sgv_set_option", "sgv_forward", "sgv_decode", "sgv_encode", "sgv_get_xhat", "sgv_get_activation",
    "sgv_backward", "sgv_set_bucket_callback", "sgv_grad_buffer", "sgv_scale_grads", "sgv_grad_norm",
    "sgv_adamw_step", "sgv_augment_collate", "sgv_dataset_convert", "sgv_dataset_sample_bytes",
    "sgv_adamw_step_range", "sgv_bucket_count", "sgv_bucket_dots", "sgv_wire_stream", "sgv_opt_stream", "sgv_adamw_bucket_async", "sgv_set_grad_payload", "sgv_grad_payload_buffer", "sgv_grad_payload_unpack", "sgv_memory_info", "sgv_recompute_bytes", "sgv_last_grad_norm", "sgv_scalars_accumulate", "sgv_scalars_read", "sgv_backward_step",
    "sgv_minmax_fit", "sgv_minmax_coeffs", "sgv_scale_convert",
    "sgv_rccl_unique_id", "sgv_rccl_probe", "sgv_rccl_comm_count", "sgv_rccl_allreduce", "sgv_rccl_comm_init", "sgv_rccl_comm_destroy", "sgv_allreduce_grads", "sgv_set_rccl", "sgv_comm_stream",
    "sgv_augment_stage", "sgv_augment_advance",
    "sgv_kernel_time", "sgv_kernel_time_reset", "sgv_kernel_time_tag", "sgv_test_gemm_nt", "sgv_test_gemm_nt_stats", "sgv_test_gemm_nt256", "sgv_test_conv_gn_fwd", "sgv_test_conv_gn_bwd", "sgv_test_gemm_tn", "sgv_test_stream_overlap", "sgv_test_occupy", "sgv_test_fake_collective",
]


class SgvConfig(C.Structure):
    _fields_ = [("latent_dim", C.c_int32), ("hierarchical_dim", C.c_int32), ("n_levels", C.c_int32),
                ("num_filter_enc", C.c_int32 * MAX_LEVELS), ("num_node", C.c_int32), ("num_time", C.c_int32),
                ("max_batch", C.c_int32), ("loss_type", C.c_int32), ("small", C.c_int32),
                ("compute_dtype", C.c_int32), ("flags", C.c_int32)]


BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t)
_lib = None


class SgvError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH):
    """dlopen libsgvae.so and declare the signatures.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise SgvError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(or `make -C simulgen-vae_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(path)
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    lib.sgv_last_error.restype = C.c_char_p
    lib.sgv_create.argtypes = [C.POINTER(SgvConfig), vp, C.POINTER(vp)]
    lib.sgv_destroy.argtypes = [vp]
    lib.sgv_param_count.argtypes = [vp]
    lib.sgv_param_info.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(i32), C.POINTER(C.c_int64),
                                   C.POINTER(i32), C.POINTER(i32)]
    lib.sgv_load_state.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.sgv_export_state.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.sgv_export_grad.argtypes = [vp, C.c_char_p, vp, C.c_size_t, C.POINTER(i32)]
    lib.sgv_export_adam.argtypes = [vp, C.c_char_p, vp, vp, C.c_size_t]
    lib.sgv_prepare.argtypes = [vp]
    lib.sgv_set_input.argtypes = [vp, vp, i32]
    lib.sgv_set_eps.argtypes = [vp, i32, vp, i32]
    lib.sgv_seed.argtypes = [vp, C.c_uint64]
    lib.sgv_set_shard.argtypes = [vp, i32, i32]
    lib.sgv_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.sgv_forward.argtypes = [vp, i32, i32, vp]
    lib.sgv_encode.argtypes = [vp, vp, vp, vp]
    lib.sgv_decode.argtypes = [vp, vp, vp, i32, i32, vp]
    lib.sgv_get_xhat.argtypes = [vp, vp]
    lib.sgv_get_activation.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.sgv_backward.argtypes = [vp, f32, f32]
    lib.sgv_set_bucket_callback.argtypes = [vp, BUCKET_CB, vp]
    lib.sgv_grad_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.sgv_scale_grads.argtypes = [vp, f32]
    lib.sgv_grad_norm.argtypes = [vp, C.POINTER(C.c_double)]
    lib.sgv_adamw_step.argtypes = [vp, f32]
    lib.sgv_backward_step.argtypes = [vp, f32, f32, f32]
    lib.sgv_adamw_step_range.argtypes = [vp, f32, i32, i32, i32, i32]
    lib.sgv_bucket_count.argtypes = [vp]
    lib.sgv_bucket_dots.argtypes = [vp, i32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.sgv_opt_stream.argtypes = [vp, C.POINTER(vp)]
    lib.sgv_wire_stream.argtypes = [vp, C.POINTER(vp)]
    lib.sgv_comm_stream.argtypes = [vp, C.POINTER(vp)]
    lib.sgv_adamw_bucket_async.argtypes = [vp, f32, i32]
    lib.sgv_set_grad_payload.argtypes = [vp, C.c_int]
    lib.sgv_grad_payload_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.sgv_grad_payload_unpack.argtypes = [vp]
    lib.sgv_rccl_unique_id.argtypes = [vp]
    lib.sgv_rccl_comm_init.argtypes = [C.POINTER(vp), i32, vp, i32]
    lib.sgv_rccl_comm_destroy.argtypes = [vp]
    lib.sgv_rccl_probe.argtypes = []
    lib.sgv_rccl_comm_count.argtypes = [vp, C.POINTER(i32)]
    lib.sgv_rccl_allreduce.argtypes = [vp, vp, C.c_size_t, i32, vp]
    lib.sgv_test_fake_collective.argtypes = [f32, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    lib.sgv_test_stream_overlap.argtypes = [vp, i32, C.POINTER(i32)]
    lib.sgv_allreduce_grads.argtypes = [vp, vp, vp]
    lib.sgv_set_rccl.argtypes = [vp, vp, vp]
    lib.sgv_last_grad_norm.argtypes = [vp, C.POINTER(C.c_double)]
    lib.sgv_memory_info.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.sgv_recompute_bytes.argtypes = [vp, C.POINTER(C.c_size_t)]
    lib.sgv_scalars_accumulate.argtypes = [vp]
    lib.sgv_scalars_read.argtypes = [vp, C.POINTER(C.c_double), i32]
    lib.sgv_augment_collate.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    lib.sgv_augment_stage.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    lib.sgv_augment_advance.argtypes = [vp]
    lib.sgv_dataset_convert.argtypes = [vp, vp, vp, i32]
    lib.sgv_dataset_sample_bytes.argtypes = [vp]
    lib.sgv_dataset_sample_bytes.restype = C.c_size_t
    lib.sgv_kernel_time.argtypes = [vp, C.c_char_p, C.POINTER(f32), C.POINTER(i32)]
    lib.sgv_kernel_time_reset.argtypes = [vp, i32]
    lib.sgv_minmax_fit.argtypes = [vp, C.c_long, i32, vp, vp, i32, vp]
    lib.sgv_minmax_coeffs.argtypes = [vp, vp, i32, f32, f32, vp, vp, vp]
    lib.sgv_scale_convert.argtypes = [i32, vp, vp, vp, vp, C.c_long, i32, vp]
    lib.sgv_kernel_time_tag.argtypes = [vp, i32, C.c_char_p, C.c_size_t, C.POINTER(f32), C.POINTER(i32)]
    lib.sgv_test_gemm_nt.argtypes = [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sgv_test_gemm_nt_stats.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.sgv_test_gemm_nt256.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.sgv_test_occupy.argtypes = [vp, i32, i32, i32, C.c_longlong]
    lib.sgv_test_gemm_tn.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.sgv_test_conv_gn_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, f32, vp]
    lib.sgv_test_conv_gn_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    _lib = lib
    return lib


def _check(lib, rc: int, what: str):
    if rc != 0:
        raise SgvError(f"{what} failed ({rc}): {lib.sgv_last_error().decode()}")


class Engine:
    """One VAE engine on the current CUDA(HIP) device and torch stream."""

    def __init__(self, cfg: VAEConfig, max_batch: int, compute_dtype: str = "bf16", flags: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise SgvError("no MI355X visible (torch.cuda.is_available() is False): libsgvae has no CPU fallback")
        if list(cfg.num_filter_dec) != list(cfg.num_filter_enc)[::-1]:
            raise SgvError("num_filter_dec must be num_filter_enc reversed (reference SimulGen-VAE.py:219)")
        self.torch = torch
        self.lib = load_library()
        self.cfg = cfg
        self.max_batch = int(max_batch)
        self.compute_dtype = compute_dtype
        c = SgvConfig()
        c.latent_dim, c.hierarchical_dim = cfg.latent_dim, cfg.hierarchical_dim
        c.n_levels = len(cfg.num_filter_enc)
        for i, v in enumerate(cfg.num_filter_enc):
            c.num_filter_enc[i] = v
        c.num_node, c.num_time, c.max_batch = cfg.num_node, cfg.num_time, self.max_batch
        c.loss_type = LOSS_IDS[cfg.lossfun]
        c.small = 1 if cfg.small else 0
        c.compute_dtype = DTYPES[compute_dtype]
        c.flags = flags
        self.stream = torch.cuda.current_stream().cuda_stream
        h = C.c_void_p()
        _check(self.lib, self.lib.sgv_create(C.byref(c), C.c_void_p(self.stream), C.byref(h)), "sgv_create")
        self.h = h
        self.spec = param_spec(cfg)
        self.n_kl = len(cfg.num_filter_enc) - 1
        self.batch = 0
        self._cb = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.sgv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ----
    def param_info(self):
        out = []
        n = self.lib.sgv_param_count(self.h)
        for i in range(n):
            name, nd, kind, hg = C.c_char_p(), C.c_int(), C.c_int(), C.c_int()
            shape = (C.c_int64 * 4)()
            _check(self.lib, self.lib.sgv_param_info(self.h, i, C.byref(name), C.byref(nd), shape, C.byref(kind),
                                                     C.byref(hg)), "sgv_param_info")
            out.append((name.value.decode(), tuple(shape[j] for j in range(nd.value)), kind.value, bool(hg.value)))
        return out

    def load_state(self, state: Dict[str, np.ndarray], partial: bool = False):
        """nn.Module.load_state_dict: every key of the reference state_dict (partial=True: only the keys given,
        e.g. to restore the spectral-norm u/v vectors)."""
        for e in self.spec:
            if partial and e.name not in state:
                continue
            a = np.ascontiguousarray(state[e.name], dtype=np.float32)
            if a.shape != tuple(e.shape):
                raise SgvError(f"shape mismatch for {e.name}: {a.shape} vs {e.shape}")
            _check(self.lib, self.lib.sgv_load_state(self.h, e.name.encode(), a.ctypes.data_as(C.c_void_p), a.size),
                   f"sgv_load_state({e.name})")
        _check(self.lib, self.lib.sgv_prepare(self.h), "sgv_prepare")

    def state_dict(self) -> Dict[str, np.ndarray]:
        out = {}
        for e in self.spec:
            a = np.empty(e.shape, dtype=np.float32)
            _check(self.lib, self.lib.sgv_export_state(self.h, e.name.encode(), a.ctypes.data_as(C.c_void_p), a.size),
                   f"sgv_export_state({e.name})")
            out[e.name] = a
        return out

    def grad(self, name: str) -> Optional[np.ndarray]:
        e = next(x for x in self.spec if x.name == name)
        a = np.empty(e.shape, dtype=np.float32)
        none = C.c_int()
        _check(self.lib, self.lib.sgv_export_grad(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size,
                                                  C.byref(none)), f"sgv_export_grad({name})")
        return None if none.value else a

    def adam_state(self, name: str):
        e = next(x for x in self.spec if x.name == name)
        m = np.empty(e.shape, dtype=np.float32)
        v = np.empty(e.shape, dtype=np.float32)
        _check(self.lib, self.lib.sgv_export_adam(self.h, name.encode(), m.ctypes.data_as(C.c_void_p),
                                                  v.ctypes.data_as(C.c_void_p), m.size), "sgv_export_adam")
        return m, v

    # ---- step ----
    def set_option(self, key: str, value: int):
        _check(self.lib, self.lib.sgv_set_option(self.h, key.encode(), int(value)), "sgv_set_option")

    def seed(self, seed: int):
        _check(self.lib, self.lib.sgv_seed(self.h, C.c_uint64(seed)), "sgv_seed")

    def set_shard(self, rank: int, world: int):
        """Sample b of this engine's batch is sample b * world + rank of the global batch: the engine's noise draws are keyed by
        that global row (include/sgvae.h: sgv_set_shard), so the same seed on every rank gives world-size-invariant noise."""
        _check(self.lib, self.lib.sgv_set_shard(self.h, int(rank), int(world)), "sgv_set_shard")

    def set_input(self, x):
        """x: torch float32 CUDA tensor [B, num_node, num_time] (reference layout)."""
        t = self.torch
        assert x.is_cuda and x.dtype == t.float32 and x.is_contiguous()
        B = x.shape[0]
        assert tuple(x.shape[1:]) == (self.cfg.num_node, self.cfg.num_time), x.shape
        _check(self.lib, self.lib.sgv_set_input(self.h, C.c_void_p(x.data_ptr()), B), "sgv_set_input")
        self.batch = B

    def set_eps(self, eps_list):
        for site, e in enumerate(eps_list):
            assert e.is_cuda and e.dtype == self.torch.float32 and e.is_contiguous()
            _check(self.lib, self.lib.sgv_set_eps(self.h, site, C.c_void_p(e.data_ptr()), e.shape[0]), "sgv_set_eps")

    def forward(self, train: bool = True, fix: bool = False, sync: bool = True):
        buf = (C.c_float * MAX_SCALARS)()
        _check(self.lib, self.lib.sgv_forward(self.h, int(train), int(fix), buf if sync else None), "sgv_forward")
        if not sync:
            return None
        n = self.n_kl
        return dict(recon=buf[0], kls=[buf[1 + i] for i in range(n)], mse=buf[1 + n])

    def decode(self, z, xs, fix: bool = False):
        """Decoder.forward(z, xs, mode): z torch fp32 CUDA [B,latent]; xs list of [B,hier] in Encoder order."""
        t = self.torch
        B = z.shape[0]
        xs_t = t.stack([x.to(t.float32) for x in xs]).contiguous().cuda()
        z = z.to(t.float32).contiguous().cuda()
        buf = (C.c_float * MAX_SCALARS)()
        _check(self.lib, self.lib.sgv_decode(self.h, C.c_void_p(z.data_ptr()), C.c_void_p(xs_t.data_ptr()), B, int(fix), buf),
               "sgv_decode")
        self.batch = B
        return [buf[2 + i] for i in range(self.n_kl - 1)]

    def encode(self):
        B, Z, Hd = self.batch, self.cfg.latent_dim, self.cfg.hierarchical_dim
        mu = np.empty((B, Z), np.float32)
        lv = np.empty((B, Z), np.float32)
        xs = np.empty((self.n_kl, B, Hd), np.float32)
        _check(self.lib, self.lib.sgv_encode(self.h, mu.ctypes.data_as(C.c_void_p), lv.ctypes.data_as(C.c_void_p),
                                             xs.ctypes.data_as(C.c_void_p)), "sgv_encode")
        return mu, lv, [xs[i] for i in range(self.n_kl)]

    def xhat(self):
        t = self.torch
        out = t.empty((self.batch, self.cfg.num_node, self.cfg.num_time), dtype=t.float32, device="cuda")
        _check(self.lib, self.lib.sgv_get_xhat(self.h, C.c_void_p(out.data_ptr())), "sgv_get_xhat")
        return out

    def activation(self, name: str, shape) -> np.ndarray:
        a = np.empty(shape, dtype=np.float32)
        _check(self.lib, self.lib.sgv_get_activation(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size),
               f"sgv_get_activation({name})")
        return a

    def backward(self, alpha: float, beta: float):
        _check(self.lib, self.lib.sgv_backward(self.h, float(alpha), float(beta)), "sgv_backward")

    def backward_step(self, alpha: float, beta: float, lr: float):
        """backward + AdamW with the optimizer of finished buckets overlapped under the rest of backward."""
        _check(self.lib, self.lib.sgv_backward_step(self.h, float(alpha), float(beta), float(lr)), "sgv_backward_step")

    def grad_norm(self) -> float:
        d = C.c_double()
        _check(self.lib, self.lib.sgv_grad_norm(self.h, C.byref(d)), "sgv_grad_norm")
        return d.value

    def adamw_step(self, lr: float):
        _check(self.lib, self.lib.sgv_adamw_step(self.h, float(lr)), "sgv_adamw_step")

    def adamw_step_range(self, lr: float, bucket_lo: int, bucket_hi: int, first: bool, last: bool):
        _check(self.lib, self.lib.sgv_adamw_step_range(self.h, float(lr), int(bucket_lo), int(bucket_hi), int(first), int(last)),
               "sgv_adamw_step_range")

    def bucket_count(self) -> int:
        return int(self.lib.sgv_bucket_count(self.h))

    def bucket_dots(self, bucket: int):
        """(offset, count) in the fp32 gradient arena of the <G,W> scalars of weight bucket `bucket`'s conv layers."""
        off, cnt = C.c_size_t(), C.c_size_t()
        _check(self.lib, self.lib.sgv_bucket_dots(self.h, int(bucket), C.byref(off), C.byref(cnt)), "sgv_bucket_dots")
        return off.value, cnt.value

    def comm_stream(self) -> int:
        """Raw HIP stream for set_rccl that the engine placed on a hardware queue of its own (include/sgvae.h: sgv_comm_stream)."""
        p = C.c_void_p()
        _check(self.lib, self.lib.sgv_comm_stream(self.h, C.byref(p)), "sgv_comm_stream")
        return p.value

    def wire_stream(self) -> int:
        """Raw HIP stream released buckets are complete (and packed) on once option "wire_stream" is set."""
        p = C.c_void_p()
        _check(self.lib, self.lib.sgv_wire_stream(self.h, C.byref(p)), "sgv_wire_stream")
        return p.value

    def opt_stream(self) -> int:
        """Raw HIP stream the ahead-of-step bucket updates run on (wrap with torch.cuda.ExternalStream)."""
        p = C.c_void_p()
        _check(self.lib, self.lib.sgv_opt_stream(self.h, C.byref(p)), "sgv_opt_stream")
        return p.value

    def adamw_bucket_async(self, lr: float, bucket: int):
        """AdamW of one weight bucket's conv weights on the optimizer stream (include/sgvae.h: the caller has made that stream
        wait for the bucket's and its <G,W> scalars' all-reduce); the closing adamw_step_range calls skip it."""
        _check(self.lib, self.lib.sgv_adamw_bucket_async(self.h, float(lr), int(bucket)), "sgv_adamw_bucket_async")

    def last_grad_norm(self) -> float:
        d = C.c_double()
        _check(self.lib, self.lib.sgv_last_grad_norm(self.h, C.byref(d)), "sgv_last_grad_norm")
        return d.value

    def memory_info(self):
        """dict of device bytes: params, grads, adam, copies, activations, workspaces."""
        buf = (C.c_size_t * 6)()
        _check(self.lib, self.lib.sgv_memory_info(self.h, buf), "sgv_memory_info")
        return dict(zip(("params", "grads", "adam", "copies", "activations", "workspaces"), [int(v) for v in buf]))

    def recompute_bytes(self) -> int:
        """Bytes of the activation maps the last backward regenerated under option "recompute_activations" (include/sgvae.h)."""
        n = C.c_size_t()
        _check(self.lib, self.lib.sgv_recompute_bytes(self.h, C.byref(n)), "sgv_recompute_bytes")
        return int(n.value)

    def accumulate_scalars(self):
        """Add the scalars of the step just enqueued to the device-side epoch accumulator (no host sync)."""
        _check(self.lib, self.lib.sgv_scalars_accumulate(self.h), "sgv_scalars_accumulate")

    def read_accumulated(self, reset: bool = True):
        """dict(recon, kls, mse, grad_norm, steps): sums over the accumulated steps (one host sync)."""
        buf = (C.c_double * 16)()
        _check(self.lib, self.lib.sgv_scalars_read(self.h, buf, int(reset)), "sgv_scalars_read")
        return dict(recon=buf[0], kls=[buf[1 + i] for i in range(self.n_kl)], mse=buf[8], grad_norm=buf[9], steps=int(buf[10]))

    def grad_buffer(self):
        """(device pointer, element count) of the flat fp32 gradient arena."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib, self.lib.sgv_grad_buffer(self.h, C.byref(p), C.byref(n)), "sgv_grad_buffer")
        return p.value, n.value

    def set_grad_payload(self, dtype: str):
        """Wire format of the data-parallel weight buckets: "f32" (the arena itself) or "bf16" (include/sgvae.h: sgv_set_grad_payload)."""
        _check(self.lib, self.lib.sgv_set_grad_payload(self.h, DTYPES[dtype]), "sgv_set_grad_payload")

    def grad_payload_buffer(self):
        """(device pointer, element count) of the bf16 payload buffer (same element offsets as the fp32 arena)."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.lib, self.lib.sgv_grad_payload_buffer(self.h, C.byref(p), C.byref(n)), "sgv_grad_payload_buffer")
        return p.value, n.value

    def grad_payload_unpack(self):
        _check(self.lib, self.lib.sgv_grad_payload_unpack(self.h), "sgv_grad_payload_unpack")

    def scale_grads(self, f: float):
        _check(self.lib, self.lib.sgv_scale_grads(self.h, float(f)), "sgv_scale_grads")

    def set_bucket_callback(self, fn):
        """fn(bucket, offset_elems, count_elems) is called from inside backward()."""
        if fn is None:
            self._cb = BUCKET_CB(0)
        else:
            self._cb = BUCKET_CB(lambda user, b, off, cnt: fn(b, off, cnt))
        _check(self.lib, self.lib.sgv_set_bucket_callback(self.h, self._cb, None), "sgv_set_bucket_callback")

    def set_rccl(self, comm, comm_stream):
        """Register an RCCL communicator (sgv_rccl_comm_init) + communication stream: backward() then issues the bucket
        all-reduces itself and adamw_step() / backward_step() wait for them bucket by bucket.  comm=None unregisters."""
        _check(self.lib, self.lib.sgv_set_rccl(self.h, C.c_void_p(comm), C.c_void_p(comm_stream) if comm else None), "sgv_set_rccl")

    def stream_overlaps(self):
        """Probe results for the engine's auxiliary streams (second lane, weight-gradient side stream, optimizer stream,
        communication stream): 1 = its kernels run beside the main stream's, 0 = it shares the main stream's hardware queue (the
        overlap it exists for is lost), -1 = the stream does not exist.  Creates the optimizer / communication streams."""
        out = {}
        for which, name in enumerate(("lane", "side", "opt", "comm")):
            v = C.c_int(-2)
            _check(self.lib, self.lib.sgv_test_stream_overlap(self.h, which, C.byref(v)), "sgv_test_stream_overlap")
            out[name] = v.value
        return out

    def stream_overlaps_existing(self):
        """stream_overlaps() for the streams a single-GPU engine has (lane, side) -- creates nothing."""
        out = {}
        for which, name in enumerate(("lane", "side")):
            v = C.c_int(-2)
            _check(self.lib, self.lib.sgv_test_stream_overlap(self.h, which, C.byref(v)), "sgv_test_stream_overlap")
            out[name] = v.value
        return out

    def allreduce_grads(self, comm, comm_stream=None):
        """Mean all-reduce of the whole gradient arena (stream-ordered, not overlapped)."""
        _check(self.lib, self.lib.sgv_allreduce_grads(self.h, C.c_void_p(comm), C.c_void_p(comm_stream) if comm_stream else None),
               "sgv_allreduce_grads")

    # ---- data ----
    def sample_bytes(self) -> int:
        return self.lib.sgv_dataset_sample_bytes(self.h)

    def dataset_convert(self, src, dst, count: int):
        _check(self.lib, self.lib.sgv_dataset_convert(self.h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                                      count), "sgv_dataset_convert")

    def augment_collate(self, dataset, idx, noise_seed, scale, mix_idx, lam):
        idx = np.ascontiguousarray(idx, np.int32)
        ns = np.ascontiguousarray(noise_seed, np.uint64)
        sc = np.ascontiguousarray(scale, np.float32)
        mi = np.ascontiguousarray(mix_idx, np.int32)
        lm = np.ascontiguousarray(lam, np.float32)
        B = len(idx)
        _check(self.lib, self.lib.sgv_augment_collate(self.h, C.c_void_p(dataset.data_ptr()), B,
                                                      idx.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p),
                                                      sc.ctypes.data_as(C.c_void_p), mi.ctypes.data_as(C.c_void_p),
                                                      lm.ctypes.data_as(C.c_void_p)), "sgv_augment_collate")
        self.batch = B

    def augment_stage(self, dataset, idx, noise_seed, scale, mix_idx, lam):
        """Prefetch: enqueue the NEXT batch's augmentation into the spare input buffer (runs beside the current step)."""
        idx = np.ascontiguousarray(idx, np.int32)
        ns = np.ascontiguousarray(noise_seed, np.uint64)
        sc = np.ascontiguousarray(scale, np.float32)
        mi = np.ascontiguousarray(mix_idx, np.int32)
        lm = np.ascontiguousarray(lam, np.float32)
        _check(self.lib, self.lib.sgv_augment_stage(self.h, C.c_void_p(dataset.data_ptr()), len(idx),
                                                    idx.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p),
                                                    sc.ctypes.data_as(C.c_void_p), mi.ctypes.data_as(C.c_void_p),
                                                    lm.ctypes.data_as(C.c_void_p)), "sgv_augment_stage")
        self._staged_batch = len(idx)

    def augment_advance(self):
        """The staged batch becomes the current input."""
        _check(self.lib, self.lib.sgv_augment_advance(self.h), "sgv_augment_advance")
        self.batch = self._staged_batch

    # ---- profiling ----
    def kernel_time_reset(self, enable: bool):
        _check(self.lib, self.lib.sgv_kernel_time_reset(self.h, int(enable)), "sgv_kernel_time_reset")

    def kernel_time_tags(self):
        """[(tag, total_ms, launches)] for every timer tag (per layer and shape after kernel_time_reset(2))."""
        out, i = [], 0
        buf = C.create_string_buffer(256)
        while True:
            ms, n = C.c_float(), C.c_int()
            if self.lib.sgv_kernel_time_tag(self.h, i, buf, 256, C.byref(ms), C.byref(n)) != 0:
                return out
            if n.value:
                out.append((buf.value.decode(), ms.value, n.value))
            i += 1

    def kernel_time(self, which: str):
        ms, n = C.c_float(), C.c_int()
        _check(self.lib, self.lib.sgv_kernel_time(self.h, which.encode(), C.byref(ms), C.byref(n)), "sgv_kernel_time")
        return ms.value, n.value
