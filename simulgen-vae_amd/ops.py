"""ctypes binding of the operator-level C ABI (include/sgvae_ops.h) used by the latent-conditioner mirror.

Every function takes / returns torch CUDA tensors (device memory + the current stream are the only things torch
is used for here); feature maps are channels-last [B, H, W, C] in the compute dtype, small tensors fp32.
There is no CPU fallback: the library must be built and a GPU present."""
from __future__ import annotations

import ctypes as C

import torch

from .engine import DTYPES, SgvError, load_library  # noqa: F401  (SgvError re-exported)

OPS_SYMBOLS = [
    "sgv_op_conv_out_shape", "sgv_op_im2col", "sgv_op_col2im", "sgv_op_conv2d_nt", "sgv_op_conv2d_tn", "sgv_op_gemm_nt", "sgv_op_gemm_nt_add_s2", "sgv_op_gemm_tn", "sgv_op_gemm_tn_splitk", "sgv_op_matvec_t", "sgv_op_gn_fwd", "sgv_op_gn_tail", "sgv_op_gn_apply", "sgv_op_stem_conv_fwd", "sgv_op_stem_conv_dw", "sgv_op_gn_relu_maxpool_fwd", "sgv_op_stem_conv_workspace_floats",
    "sgv_op_gn_workspace_floats", "sgv_op_gn_bwd", "sgv_op_gn_bwd_set", "sgv_op_maxpool_fwd", "sgv_op_maxpool_bwd", "sgv_op_add_relu_fwd",
    "sgv_op_relu_bwd", "sgv_op_add", "sgv_op_avgpool_fwd", "sgv_op_avgpool_bwd", "sgv_op_chan_scale_fwd",
    "sgv_op_chan_scale_bwd", "sgv_op_linear_fwd", "sgv_op_act_fwd", "sgv_op_act_bwd", "sgv_op_linear_bwd", "sgv_op_layernorm_fwd",
    "sgv_op_layernorm_bwd", "sgv_op_batchnorm_fwd", "sgv_op_batchnorm_bwd", "sgv_op_mask_scale", "sgv_op_addf",
    "sgv_op_mse", "sgv_op_loss_value", "sgv_op_cols_sub_div", "sgv_op_multi_copy", "sgv_op_transpose", "sgv_op_l2_normalize", "sgv_op_dot", "sgv_op_sn_grad", "sgv_op_conv_weight_pack",
    "sgv_op_conv_weight_unpack", "sgv_op_sumsq", "sgv_op_clip_coef", "sgv_op_adamw", "sgv_op_flip_roll", "sgv_op_affine_sample",
    "sgv_op_mixup_rows", "sgv_pset_create", "sgv_pset_destroy", "sgv_pset_power_iteration", "sgv_pset_sigma", "sgv_pset_step",
]
ACT_NONE, ACT_RELU_GN = 0, 3            # GroupNorm activation ids (ew.hip)
LIN_NONE, LIN_RELU, LIN_SIGMOID = 0, 1, 2

_lib = None


def lib():
    global _lib
    if _lib is None:
        l = load_library()
        vp, i, f, lg = C.c_void_p, C.c_int, C.c_float, C.c_long
        sig = {
            "sgv_op_conv_out_shape": [i] * 7 + [C.POINTER(i)] * 3,
            "sgv_op_im2col": [i, vp, vp] + [i] * 8 + [vp],
            "sgv_op_col2im": [i, vp, vp] + [i] * 8 + [vp],
            "sgv_op_gemm_nt": [i, vp, vp, vp, vp, vp, vp, i, i, i, i, vp],
            "sgv_op_gemm_nt_add_s2": [i, vp, vp, vp, vp, vp, i, i, i, i, i, vp],
            "sgv_op_conv2d_nt": [i, vp, vp, vp, vp] + [i] * 9 + [C.c_long, C.c_long, i, vp],
            "sgv_op_gemm_tn": [i, vp, vp, vp, i, i, i, vp, i, vp],
            "sgv_op_conv2d_tn": [i, vp, vp, vp] + [i] * 9 + [vp, i, vp],
            "sgv_op_gemm_tn_splitk": [i, i, i, i],
            "sgv_op_matvec_t": [vp, vp, vp, i, i, vp],
            "sgv_op_gn_fwd": [i, i, vp, vp, i, i, i, i, vp, vp, vp, vp, vp],
            "sgv_op_gn_workspace_floats": [i, i, i],
            "sgv_op_gn_apply": [i, i, vp, vp, i, i, i, i, vp, vp, vp, vp],
            "sgv_op_stem_conv_workspace_floats": [i, i, i, i],
            "sgv_op_stem_conv_fwd": [vp] * 6 + [i] * 8 + [vp],
            "sgv_op_stem_conv_dw": [vp] * 4 + [i] * 7 + [vp],
            "sgv_op_gn_tail": [i] + [vp] * 10 + [i, i, i, i, vp, vp],
            "sgv_op_gn_bwd": [i, i, vp, vp, vp, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp],
            "sgv_op_gn_bwd_set": [i, i, vp, vp, vp, i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp],
            "sgv_op_maxpool_fwd": [i, vp, vp, vp, i, i, i, i, vp],
            "sgv_op_gn_relu_maxpool_fwd": [i, vp, vp, vp, vp, i, vp, vp, vp, i, i, i, i, vp],
            "sgv_op_maxpool_bwd": [i, vp, vp, vp, i, i, i, i, vp],
            "sgv_op_add_relu_fwd": [i, vp, vp, vp, lg, vp],
            "sgv_op_relu_bwd": [i, vp, vp, vp, lg, vp],
            "sgv_op_add": [i, vp, vp, vp, lg, vp],
            "sgv_op_avgpool_fwd": [i, vp, vp, i, i, i, vp],
            "sgv_op_avgpool_bwd": [i, vp, vp, i, i, i, i, vp],
            "sgv_op_chan_scale_fwd": [i, vp, vp, vp, i, i, i, vp],
            "sgv_op_chan_scale_bwd": [i, vp, vp, vp, vp, vp, i, i, i, vp],
            "sgv_op_linear_fwd": [vp, vp, vp, vp, vp, i, i, i, i, vp],
            "sgv_op_act_fwd": [vp, vp, lg, i, vp],
            "sgv_op_act_bwd": [vp, vp, vp, lg, i, vp],
            "sgv_op_linear_bwd": [vp, vp, vp, vp, vp, i, vp, vp, i, i, i, vp],
            "sgv_op_layernorm_fwd": [vp, vp, vp, vp, vp, i, i, vp],
            "sgv_op_layernorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i, i, vp],
            "sgv_op_batchnorm_fwd": [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp],
            "sgv_op_batchnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp],
            "sgv_op_mask_scale": [vp, vp, f, vp, lg, vp],
            "sgv_op_addf": [vp, vp, vp, lg, vp],
            "sgv_op_mse": [vp, vp, vp, vp, f, lg, vp],
            "sgv_op_loss_value": [i, vp, vp, vp, f, lg, vp],
            "sgv_op_multi_copy": [vp, i, vp],
            "sgv_op_cols_sub_div": [vp, vp, vp, vp, lg, i, vp],
            "sgv_op_transpose": [i, i, vp, vp, i, i, i, vp],
            "sgv_op_l2_normalize": [vp, vp, lg, f, vp],
            "sgv_op_dot": [vp, vp, vp, lg, vp],
            "sgv_op_sn_grad": [vp, vp, vp, vp, vp, vp, i, i, vp],
            "sgv_op_conv_weight_pack": [i, vp, vp, i, i, i, i, vp],
            "sgv_op_conv_weight_unpack": [vp, vp, i, i, i, i, vp],
            "sgv_op_sumsq": [vp, lg, vp, vp],
            "sgv_op_clip_coef": [vp, f, vp, vp],
            "sgv_op_adamw": [vp, vp, vp, vp, lg, f, f, f, f, f, i, vp, vp],
            "sgv_op_flip_roll": [vp, vp, i, i, i, vp, vp, vp, vp],
            "sgv_op_affine_sample": [vp, vp, i, i, i, vp, vp],
            "sgv_op_mixup_rows": [vp, vp, f, vp, i, lg, vp],
        }
        sig["sgv_pset_create"] = [vp, i, vp]
        sig["sgv_pset_destroy"] = [vp]
        sig["sgv_pset_power_iteration"] = [vp, i, vp]
        sig["sgv_pset_sigma"] = [vp, i]
        sig["sgv_pset_step"] = [vp, f, f, f, vp, vp]
        for name, args in sig.items():
            getattr(l, name).argtypes = args
        l.sgv_pset_sigma.restype = C.c_void_p
        l.sgv_op_gn_workspace_floats.restype = C.c_size_t
        l.sgv_op_stem_conv_workspace_floats.restype = C.c_size_t
        _lib = l
    return _lib


def _p(t):
    if t is None:
        return None
    return C.c_void_p(t) if isinstance(t, int) else C.c_void_p(t.data_ptr())        # raw device address or tensor


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ck(rc, what):
    if rc != 0:
        raise SgvError(f"{what} failed ({rc}): {lib().sgv_last_error().decode()}")


def tdtype(dtype: str):
    return torch.bfloat16 if DTYPES[dtype] == 1 else torch.float32


def conv_out_shape(H, W, Cin, KH, KW, stride, pad):
    ho, wo, kp = C.c_int(), C.c_int(), C.c_int()
    lib().sgv_op_conv_out_shape(H, W, Cin, KH, KW, stride, pad, C.byref(ho), C.byref(wo), C.byref(kp))
    return ho.value, wo.value, kp.value


def _d(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def im2col(x, KH, KW, stride, pad):
    B, H, W, Cin = x.shape
    Ho, Wo, Kp = conv_out_shape(H, W, Cin, KH, KW, stride, pad)
    col = torch.empty((B * Ho * Wo, Kp), dtype=x.dtype, device=x.device)
    _ck(lib().sgv_op_im2col(_d(x), _p(x), _p(col), B, H, W, Cin, KH, KW, stride, pad, _stream()), "sgv_op_im2col")
    return col, Ho, Wo


def col2im(dcol, shape, KH, KW, stride, pad):
    B, H, W, Cin = shape
    dx = torch.empty(shape, dtype=dcol.dtype, device=dcol.device)
    _ck(lib().sgv_op_col2im(_d(dcol), _p(dcol), _p(dx), B, H, W, Cin, KH, KW, stride, pad, _stream()), "sgv_op_col2im")
    return dx


# bench.py --workload lc: per-class GEMM timing with events on torch's current stream (the stream these operators run on).
# None = off; a list collects (class, flops, start event, end event).
GEMM_TIMING = None


def _timed(cls, flops, fn):
    if GEMM_TIMING is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    GEMM_TIMING.append((cls, flops, e0, e1))
    return r


def gemm_nt(A, W, bias=None, scale=None, addend=None, out_f32=False):
    """A [M, K] . W [N, K]^T -> [M, N]."""
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty((M, N), dtype=torch.float32 if out_f32 else A.dtype, device=A.device)
    _timed("gemm_nt", 2.0 * M * N * K, lambda: _ck(lib().sgv_op_gemm_nt(_d(A), _p(A), _p(W), _p(out), _p(bias), _p(scale), _p(addend), M, N, K,
                                                                         int(out_f32), _stream()), "sgv_op_gemm_nt"))
    return out


def gemm_nt_add_s2(A, W, addend_half, H, Wd, scale=None):
    """A [B*H*W, K] . W [N, K]^T + addend_half [B, ceil(H/2), ceil(W/2), N] added at the even pixels (sgv_op_gemm_nt_add_s2) -> [M, N]."""
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    _timed("gemm_nt", 2.0 * M * N * K, lambda: _ck(lib().sgv_op_gemm_nt_add_s2(_d(A), _p(A), _p(W), _p(out), _p(scale), _p(addend_half), M, N, K, H, Wd,
                                                                                _stream()), "sgv_op_gemm_nt_add_s2"))
    return out


def conv2d_nt(x, W, N, KH, KW, stride, pad, ldw, w_tap_stride, flip=False, scale=None):
    """Implicit-GEMM convolution of a channels-last batch x [B, H, W, Cin] (include/sgvae_ops.h: sgv_op_conv2d_nt): tap t's
    [N, Cin] weight matrix lies at W + t*w_tap_stride (elements) with row pitch ldw.  -> [B, Ho, Wo, N]."""
    B, H, Wd, Cin = x.shape
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (Wd + 2 * pad - KW) // stride + 1
    out = torch.empty((B, Ho, Wo, N), dtype=x.dtype, device=x.device)
    _timed("gemm_nt", 2.0 * B * Ho * Wo * N * Cin * KH * KW,
           lambda: _ck(lib().sgv_op_conv2d_nt(_d(x), _p(x), _p(W), _p(out), _p(scale), B, H, Wd, Cin, N, KH, KW, stride, pad, ldw, w_tap_stride,
                                              int(bool(flip)), _stream()), "sgv_op_conv2d_nt"))
    return out


def gemm_tn(A, Bm):
    """A [M, N1]^T . Bm [M, N2] -> fp32 [N1, N2]."""
    M, N1 = A.shape
    N2 = Bm.shape[1]
    out = torch.empty((N1, N2), dtype=torch.float32, device=A.device)
    sk = int(lib().sgv_op_gemm_tn_splitk(_d(A), M, N1, N2))
    slabs = torch.empty((sk, N1, N2), dtype=torch.float32, device=A.device) if sk > 1 else None
    _timed("gemm_tn", 2.0 * M * N1 * N2, lambda: _ck(lib().sgv_op_gemm_tn(_d(A), _p(A), _p(Bm), _p(out), M, N1, N2, _p(slabs), sk, _stream()),
                                                     "sgv_op_gemm_tn"))
    return out


def conv2d_tn(dy, x, KH, KW, stride, pad):
    """Weight gradient of the convolution x [B, H, W, Cin] -> dy [B, Ho, Wo, N1] without the im2col matrix
    (sgv_op_conv2d_tn) -> fp32 [N1, KH*KW*Cin] in the packed weight layout."""
    B, H, Wd, Cin = x.shape
    N1 = dy.shape[-1]
    M, N2 = dy.numel() // N1, KH * KW * Cin
    out = torch.empty((N1, N2), dtype=torch.float32, device=x.device)
    sk = int(lib().sgv_op_gemm_tn_splitk(_d(x), M, N1, N2))
    slabs = torch.empty((sk, N1, N2), dtype=torch.float32, device=x.device) if sk > 1 else None
    _timed("gemm_tn", 2.0 * M * N1 * N2, lambda: _ck(lib().sgv_op_conv2d_tn(_d(x), _p(dy), _p(x), _p(out), B, H, Wd, Cin, N1, KH, KW, stride, pad,
                                                                          _p(slabs), sk, _stream()), "sgv_op_conv2d_tn"))
    return out


def gn_fwd(y, G, gamma, beta, act):
    """y [B, P, C] -> (out, sums)."""
    B, P, Cc = y.shape
    out = torch.empty_like(y)
    sums = torch.empty(B * G * 2, dtype=torch.float64, device=y.device)
    part = torch.empty(int(lib().sgv_op_gn_workspace_floats(B, P, Cc)), dtype=torch.float32, device=y.device)
    _ck(lib().sgv_op_gn_fwd(_d(y), act, _p(y), _p(out), B, P, Cc, G, _p(gamma), _p(beta), _p(sums), _p(part), _stream()), "sgv_op_gn_fwd")
    return out, sums


def stem_conv(x, wp, N, k, pad, G, scale=None):
    """One-input-channel convolution on the MFMA without an im2col matrix + the GroupNorm statistics of its output
    (sgv_op_stem_conv_fwd): x [B, H, W] bf16, wp the packed weights [N, roundup(k*k, 8)] -> (y [B, H, W, N], sums)."""
    B, H, W = x.shape
    y = torch.empty((B, H, W, N), dtype=torch.bfloat16, device=x.device)
    sums = torch.empty(B * G * 2, dtype=torch.float64, device=x.device)
    part = torch.empty(int(lib().sgv_op_stem_conv_workspace_floats(B, H, W, N)), dtype=torch.float32, device=x.device)
    _timed("gemm_nt", 2.0 * B * H * W * N * k * k,
           lambda: _ck(lib().sgv_op_stem_conv_fwd(_p(x), _p(wp), _p(scale), _p(y), _p(sums), _p(part), B, H, W, N, k, k, pad, G, _stream()),
                       "sgv_op_stem_conv_fwd"))
    return y, sums


def stem_conv_dw(x, dy, k, pad):
    """Weight gradient of the stem convolution (sgv_op_stem_conv_dw): x [B, H, W] bf16, dy [B, H, W, N] bf16 -> fp32
    [N, roundup(k*k, 8)] in the packed layout."""
    B, H, W = x.shape
    N = dy.shape[-1]
    out = torch.empty((N, (k * k + 7) // 8 * 8), dtype=torch.float32, device=x.device)
    part = torch.empty(int(lib().sgv_op_stem_conv_workspace_floats(B, H, W, N)), dtype=torch.float32, device=x.device)
    _timed("gemm_tn", 2.0 * B * H * W * N * k * k,
           lambda: _ck(lib().sgv_op_stem_conv_dw(_p(x), _p(dy), _p(out), _p(part), B, H, W, N, k, k, pad, _stream()), "sgv_op_stem_conv_dw"))
    return out


def gn_apply(y, G, gamma, beta, sums, act):
    """act(gn(y)) with the statistics given. y [B, P, C]."""
    B, P, Cc = y.shape
    out = torch.empty_like(y)
    _ck(lib().sgv_op_gn_apply(_d(y), act, _p(y), _p(out), B, P, Cc, G, _p(gamma), _p(beta), _p(sums), _stream()), "sgv_op_gn_apply")
    return out


def gn_tail(y, G, gamma, beta, y2, gamma2=None, beta2=None, cscale=None):
    """out = relu(A + gn(y)) in one pass (sgv_op_gn_tail); A = gn(y2; gamma2, beta2), or y2 * cscale[b, c] when cscale is given.
    y, y2 [B, P, C] -> (out, sums of y, sums of y2 or None)."""
    B, P, Cc = y.shape
    out = torch.empty_like(y)
    sums = torch.empty(B * G * 2, dtype=torch.float64, device=y.device)
    sums2 = None if cscale is not None else torch.empty(B * G * 2, dtype=torch.float64, device=y.device)
    part = torch.empty(int(lib().sgv_op_gn_workspace_floats(B, P, Cc)), dtype=torch.float32, device=y.device)
    _ck(lib().sgv_op_gn_tail(_d(y), _p(y), _p(gamma), _p(beta), _p(sums), _p(y2), _p(gamma2), _p(beta2), _p(sums2), _p(cscale), _p(out),
                             B, P, Cc, G, _p(part), _stream()), "sgv_op_gn_tail")
    return out, sums, sums2


def gn_bwd(y, dout, G, gamma, beta, sums, act, dgamma, dbeta, accumulate=True):
    """dy; dgamma / dbeta += (accumulate=True, the buffers must hold valid numbers) or = (accumulate=False: any buffer)."""
    B, P, Cc = y.shape
    dy = torch.empty_like(y)
    sums2 = torch.empty_like(sums)
    part = torch.empty(int(lib().sgv_op_gn_workspace_floats(B, P, Cc)), dtype=torch.float32, device=y.device)
    fn = lib().sgv_op_gn_bwd if accumulate else lib().sgv_op_gn_bwd_set
    _ck(fn(_d(y), act, _p(y), _p(dout), _p(dy), B, P, Cc, G, _p(gamma), _p(beta), _p(sums), _p(sums2), _p(part),
           _p(dgamma), _p(dbeta), _stream()), "sgv_op_gn_bwd")
    return dy


def maxpool_fwd(x):
    B, H, W, Cc = x.shape
    Ho, Wo, _ = conv_out_shape(H, W, Cc, 3, 3, 2, 1)
    y = torch.empty((B, Ho, Wo, Cc), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
    _ck(lib().sgv_op_maxpool_fwd(_d(x), _p(x), _p(y), _p(idx), B, H, W, Cc, _stream()), "sgv_op_maxpool_fwd")
    return y, idx


def gn_relu_maxpool_fwd(y, G, gamma, beta, sums):
    """maxpool(relu(gn(y))) in one pass (sgv_op_gn_relu_maxpool_fwd): y [B, H, W, C] pre-norm, sums its statistics -> (out, argmax)."""
    B, H, W, Cc = y.shape
    Ho, Wo, _ = conv_out_shape(H, W, Cc, 3, 3, 2, 1)
    out = torch.empty((B, Ho, Wo, Cc), dtype=y.dtype, device=y.device)
    idx = torch.empty((B, Ho, Wo, Cc), dtype=torch.uint8, device=y.device)
    coef = torch.empty(2 * B * Cc, dtype=torch.float32, device=y.device)
    _ck(lib().sgv_op_gn_relu_maxpool_fwd(_d(y), _p(y), _p(sums), _p(gamma), _p(beta), G, _p(out), _p(idx), _p(coef), B, H, W, Cc, _stream()),
        "sgv_op_gn_relu_maxpool_fwd")
    return out, idx


def maxpool_bwd(idx, dy, in_shape):
    B, H, W, Cc = in_shape
    dx = torch.empty(in_shape, dtype=dy.dtype, device=dy.device)
    _ck(lib().sgv_op_maxpool_bwd(_d(dy), _p(idx), _p(dy), _p(dx), B, H, W, Cc, _stream()), "sgv_op_maxpool_bwd")
    return dx


def add_relu(a, b):
    out = torch.empty_like(a)
    _ck(lib().sgv_op_add_relu_fwd(_d(a), _p(a), _p(b), _p(out), a.numel(), _stream()), "sgv_op_add_relu_fwd")
    return out


def relu_bwd(out, dout):
    d = torch.empty_like(out)
    _ck(lib().sgv_op_relu_bwd(_d(out), _p(out), _p(dout), _p(d), out.numel(), _stream()), "sgv_op_relu_bwd")
    return d


def add(a, b):
    out = torch.empty_like(a)
    _ck(lib().sgv_op_add(_d(a), _p(a), _p(b), _p(out), a.numel(), _stream()), "sgv_op_add")
    return out


def avgpool_fwd(x):
    """x [B, P, C] -> fp32 [B, C]."""
    B, P, Cc = x.shape
    y = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    _ck(lib().sgv_op_avgpool_fwd(_d(x), _p(x), _p(y), B, P, Cc, _stream()), "sgv_op_avgpool_fwd")
    return y


def avgpool_bwd(dy, dx_or_shape, dtype=None):
    """dy fp32 [B, C]; either accumulates into an existing dx [B, P, C] or creates one of `dx_or_shape`."""
    if torch.is_tensor(dx_or_shape):
        dx, acc = dx_or_shape, 1
    else:
        dx, acc = torch.empty(dx_or_shape, dtype=dtype, device=dy.device), 0
    B, P, Cc = dx.shape
    _ck(lib().sgv_op_avgpool_bwd(_d(dx), _p(dy), _p(dx), B, P, Cc, acc, _stream()), "sgv_op_avgpool_bwd")
    return dx


def chan_scale_fwd(x, s):
    B, P, Cc = x.shape
    out = torch.empty_like(x)
    _ck(lib().sgv_op_chan_scale_fwd(_d(x), _p(x), _p(s), _p(out), B, P, Cc, _stream()), "sgv_op_chan_scale_fwd")
    return out


def chan_scale_bwd(x, s, dout):
    B, P, Cc = x.shape
    dx = torch.empty_like(x)
    ds = torch.empty((B, Cc), dtype=torch.float32, device=x.device)
    _ck(lib().sgv_op_chan_scale_bwd(_d(x), _p(x), _p(s), _p(dout), _p(dx), _p(ds), B, P, Cc, _stream()), "sgv_op_chan_scale_bwd")
    return dx, ds


def linear_fwd(x, W, bias=None, scale=None, act=LIN_NONE):
    B, K = x.shape
    O = W.shape[0]
    y = torch.empty((B, O), dtype=torch.float32, device=x.device)
    _ck(lib().sgv_op_linear_fwd(_p(x), _p(W), _p(bias), _p(scale), _p(y), B, K, O, act, _stream()), "sgv_op_linear_fwd")
    return y


def act_fwd(x, act):
    y = torch.empty_like(x)
    _ck(lib().sgv_op_act_fwd(_p(x), _p(y), x.numel(), act, _stream()), "sgv_op_act_fwd")
    return y


def act_bwd(y, dy, act):
    dz = torch.empty_like(y)
    _ck(lib().sgv_op_act_bwd(_p(y), _p(dy), _p(dz), y.numel(), act, _stream()), "sgv_op_act_bwd")
    return dz


def linear_bwd(dz, x, W, scale=None, need_dx=True, has_bias=True, dx_accumulate=None, need_dw=True):
    """-> (dx = scale * dz W or None, dW = scale * dz^T x or None, db or None); need_dw=False: the input gradient only."""
    B, K = x.shape
    O = W.shape[0]
    dW = torch.empty_like(W) if need_dw else None
    db = torch.empty(O, dtype=torch.float32, device=x.device) if has_bias and need_dw else None
    dx = dx_accumulate if dx_accumulate is not None else (torch.empty_like(x) if need_dx else None)
    _ck(lib().sgv_op_linear_bwd(_p(dz), _p(x), _p(W), _p(scale), _p(dx), int(dx_accumulate is not None), _p(dW), _p(db), B, K, O,
                                _stream()), "sgv_op_linear_bwd")
    return dx, dW, db


def layernorm_fwd(x, gamma, beta):
    B, K = x.shape
    y = torch.empty_like(x)
    stat = torch.empty(2 * B, dtype=torch.float32, device=x.device)
    _ck(lib().sgv_op_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stat), B, K, _stream()), "sgv_op_layernorm_fwd")
    return y, stat


def layernorm_bwd(x, gamma, stat, dy):
    B, K = x.shape
    dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
    _ck(lib().sgv_op_layernorm_bwd(_p(x), _p(gamma), _p(stat), _p(dy), _p(dx), _p(dg), _p(db), B, K, _stream()), "sgv_op_layernorm_bwd")
    return dx, dg, db


def batchnorm_fwd(x, gamma, beta, run_mean, run_var, train):
    B, K = x.shape
    y = torch.empty_like(x)
    stat = torch.empty(2 * K, dtype=torch.float32, device=x.device)
    _ck(lib().sgv_op_batchnorm_fwd(_p(x), _p(gamma), _p(beta), _p(run_mean), _p(run_var), _p(y), _p(stat), B, K, int(train), _stream()),
        "sgv_op_batchnorm_fwd")
    return y, stat


def batchnorm_bwd(x, gamma, stat, dy, train):
    B, K = x.shape
    dx, dg, db = torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)
    _ck(lib().sgv_op_batchnorm_bwd(_p(x), _p(gamma), _p(stat), _p(dy), _p(dx), _p(dg), _p(db), B, K, int(train), _stream()),
        "sgv_op_batchnorm_bwd")
    return dx, dg, db


def mask_scale(a, mask, scale):
    out = torch.empty_like(a)
    _ck(lib().sgv_op_mask_scale(_p(a), _p(mask), float(scale), _p(out), a.numel(), _stream()), "sgv_op_mask_scale")
    return out


def addf(a, b):
    out = torch.empty_like(a)
    _ck(lib().sgv_op_addf(_p(a), _p(b), _p(out), a.numel(), _stream()), "sgv_op_addf")
    return out


def mse(pred, target, gscale=1.0, need_grad=True):
    loss = torch.empty(1, dtype=torch.float64, device=pred.device)
    dp = torch.empty_like(pred) if need_grad else None
    _ck(lib().sgv_op_mse(_p(pred), _p(target), _p(loss), _p(dp), float(gscale), pred.numel(), _stream()), "sgv_op_mse")
    return loss, dp


LOSS_KINDS = {"MSE": 0, "MAE": 1, "Huber": 2, "SmoothL1": 3}


def loss_value(kind, a, b, delta=0.1):
    """Mean-reduced loss value (fp64 [1] on the device) of nn.MSELoss / L1Loss / HuberLoss(delta) / SmoothL1Loss(beta)."""
    loss = torch.empty(1, dtype=torch.float64, device=a.device)
    _ck(lib().sgv_op_loss_value(LOSS_KINDS[kind] if isinstance(kind, str) else int(kind), _p(a), _p(b), _p(loss), float(delta),
                                a.numel(), _stream()), "sgv_op_loss_value")
    return loss


def cols_sub_div(x, col_min, col_scale):
    """MinMaxScaler.inverse_transform on fp32 [rows, cols]."""
    y = torch.empty_like(x)
    _ck(lib().sgv_op_cols_sub_div(_p(x), _p(col_min), _p(col_scale), _p(y), x.shape[0], x.shape[1], _stream()), "sgv_op_cols_sub_div")
    return y


def multi_copy(pairs):
    """pairs: [(src fp32 CUDA tensor, dst fp32 CUDA view of the same numel)]: one launch for all of them."""
    import numpy as np
    tab = np.empty((len(pairs), 3), np.int64)
    for k, (src, dst) in enumerate(pairs):
        tab[k] = (src.data_ptr(), dst.data_ptr(), src.numel())
    dev = torch.from_numpy(tab).cuda()
    _ck(lib().sgv_op_multi_copy(_p(dev), len(pairs), _stream()), "sgv_op_multi_copy")
    return dev          # keep alive until the launch is enqueued (stream-ordered allocator)


def transpose(src, dst_dtype, Bn, I, J):
    """[Bn, I, J] -> [Bn, J, I] with dtype conversion."""
    dst = torch.empty((Bn, J, I), dtype=dst_dtype, device=src.device)
    _ck(lib().sgv_op_transpose(_d(src), 1 if dst_dtype == torch.bfloat16 else 0, _p(src), _p(dst), Bn, I, J, _stream()), "sgv_op_transpose")
    return dst


# ---- parameter side ----
def l2_normalize(x, eps=1e-12):
    out = torch.empty_like(x)
    _ck(lib().sgv_op_l2_normalize(_p(x), _p(out), x.numel(), float(eps), _stream()), "sgv_op_l2_normalize")
    return out


def dot(a, b):
    """-> fp32 [4]: {a.b, 1/(a.b), scratch}"""
    out = torch.empty(4, dtype=torch.float32, device=a.device)
    _ck(lib().sgv_op_dot(_p(a), _p(b), _p(out), a.numel(), _stream()), "sgv_op_dot")
    return out


def sn_power_iteration(Wm, u, v, train):
    """Legacy spectral norm on the [rows, cols] fp32 matrix view of a weight: updates u, v in place when `train`,
    returns sigma2 = {sigma, 1/sigma} (device)."""
    rows, cols = Wm.shape
    if train:
        wtu = torch.empty(cols, dtype=torch.float32, device=Wm.device)
        _ck(lib().sgv_op_matvec_t(_p(Wm), _p(u), _p(wtu), rows, cols, _stream()), "sgv_op_matvec_t")      # W^T u
        v.copy_(l2_normalize(wtu))
        u.copy_(l2_normalize(linear_fwd(v.view(1, cols), Wm).view(-1)))
    return dot(u, linear_fwd(v.view(1, cols), Wm).view(-1))


def sn_grad(G, u, v, Wm, sigma2):
    gw = dot(G.reshape(-1), Wm.reshape(-1))
    out = torch.empty_like(G)
    rows, cols = Wm.shape
    _ck(lib().sgv_op_sn_grad(_p(G), _p(u), _p(v), _p(gw), _p(sigma2), _p(out), rows, cols, _stream()), "sgv_op_sn_grad")
    return out


def conv_weight_pack(w, dtype):
    Cout, Cin, KH, KW = w.shape
    Kp = (KH * KW * Cin + 7) // 8 * 8
    out = torch.empty((Cout, Kp), dtype=dtype, device=w.device)
    _ck(lib().sgv_op_conv_weight_pack(1 if dtype == torch.bfloat16 else 0, _p(w), _p(out), Cout, Cin, KH, KW, _stream()), "sgv_op_conv_weight_pack")
    return out


def conv_weight_unpack(packed, shape):
    Cout, Cin, KH, KW = shape
    w = torch.empty(shape, dtype=torch.float32, device=packed.device)
    _ck(lib().sgv_op_conv_weight_unpack(_p(packed), _p(w), Cout, Cin, KH, KW, _stream()), "sgv_op_conv_weight_unpack")
    return w


def sumsq(g, acc):
    _ck(lib().sgv_op_sumsq(_p(g), g.numel(), _p(acc), _stream()), "sgv_op_sumsq")


def clip_coef(sumsq_acc, max_norm):
    out = torch.empty(2, dtype=torch.float32, device=sumsq_acc.device)
    _ck(lib().sgv_op_clip_coef(_p(sumsq_acc), float(max_norm), _p(out), _stream()), "sgv_op_clip_coef")
    return out


def adamw(p, g, m, v, lr, step, weight_decay, gscale=None, betas=(0.9, 0.999), eps=1e-8):
    _ck(lib().sgv_op_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(betas[0]), float(betas[1]), float(eps),
                           float(weight_decay), int(step), _p(gscale), _stream()), "sgv_op_adamw")


# ---- input augmentation ----
def _ivec(v, device):
    return torch.as_tensor(list(v), dtype=torch.int32, device=device)


def flip_roll(x, flip, shift_x, shift_y):
    """x fp32 [B, H, W]; per-sample flags / shifts (python sequences)."""
    B, H, W = x.shape
    out = torch.empty_like(x)
    f, sx, sy = _ivec(flip, x.device), _ivec(shift_x, x.device), _ivec(shift_y, x.device)
    _ck(lib().sgv_op_flip_roll(_p(x), _p(out), B, H, W, _p(f), _p(sx), _p(sy), _stream()), "sgv_op_flip_roll")
    return out


def affine_sample(x, theta):
    """x fp32 [B, H, W]; theta fp32 [B, 2, 3] (device or host)."""
    B, H, W = x.shape
    th = torch.as_tensor(theta, dtype=torch.float32).to(x.device).contiguous()
    out = torch.empty_like(x)
    _ck(lib().sgv_op_affine_sample(_p(x), _p(out), B, H, W, _p(th), _stream()), "sgv_op_affine_sample")
    return out


def mixup_rows(x, perm, lam):
    B = x.shape[0]
    n = x.numel() // B
    out = torch.empty_like(x)
    pm = _ivec(perm, x.device)
    _ck(lib().sgv_op_mixup_rows(_p(x), _p(pm), float(lam), _p(out), B, n, _stream()), "sgv_op_mixup_rows")
    return out


# ---- parameter set ----
class PsetEntry(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("n", C.c_long), ("rows", C.c_int), ("cols", C.c_int), ("u", C.c_void_p), ("v", C.c_void_p)]


class ParamSet:
    """sgv_pset_*: multi-tensor spectral-norm power iteration + clip + AdamW over fixed (parameter, gradient) buffers.
    entries: list of dicts(p=, g=, u=None, v=None, rows=0, cols=0) of fp32 CUDA tensors (kept alive by this object)."""

    def __init__(self, entries):
        self.entries = entries
        arr = (PsetEntry * len(entries))()
        for k, e in enumerate(entries):
            arr[k].p, arr[k].g, arr[k].n = e["p"].data_ptr(), e["g"].data_ptr(), e["p"].numel()
            arr[k].rows, arr[k].cols = int(e.get("rows", 0)), int(e.get("cols", 0))
            arr[k].u = e["u"].data_ptr() if e.get("u") is not None else None
            arr[k].v = e["v"].data_ptr() if e.get("v") is not None else None
        h = C.c_void_p()
        _ck(lib().sgv_pset_create(arr, len(entries), C.byref(h)), "sgv_pset_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib().sgv_pset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def power_iteration(self, train):
        _ck(lib().sgv_pset_power_iteration(self.h, int(train), _stream()), "sgv_pset_power_iteration")

    def sigma_ptr(self, entry):
        """device address of {sigma, 1/sigma} of a normalised entry"""
        return int(lib().sgv_pset_sigma(self.h, entry))

    def step(self, lr, weight_decay, max_norm, want_norm=True):
        out = C.c_float(0.0)
        _ck(lib().sgv_pset_step(self.h, float(lr), float(weight_decay), float(max_norm), C.byref(out) if want_norm else None, _stream()),
            "sgv_pset_step")
        return out.value if want_norm else None
