"""`LatentConditionerImg` with the constructor / call surface of the reference's
modules.latent_conditioner_model_cnn.LatentConditionerImg (latent_conditioner_model_cnn.py:138-362), on the MI355X.

The layer graph and its hand-derived backward live here on the host; every tensor operation is a HIP kernel behind the
operator-level C ABI (include/sgvae_ops.h, `simulgen_vae_amd.ops`): convolutions are implicit GEMMs on the MFMA kernels of
the VAE path (no im2col matrix; SGV_LC_IMPLICIT=0 restores the im2col lowering), the tail of a residual block and the stem's
GroupNorm + ReLU + max-pool are single passes (SGV_LC_FUSED_TAIL=0: separate operators), feature maps are channels-last
[B, H, W, C] in the compute dtype, the heads are fp32.  torch supplies device
memory, the stream and the random bits of the dropout masks -- nothing else (the two data-dependent input-range lines
of the reference forward, `if x.min() < -0.1: x = (x + 1) / 2`, are kept as a torch expression on the input batch).

  m = LatentConditionerImg(latent_conditioner_filter, latent_dim_end, input_shape, latent_dim, size2,
                           latent_conditioner_data_shape, dropout_rate=0.3, use_attention=True, return_dict=False)
  latent_main, xs = m(x)                       # x: [B, H*W] flattened square images, as in the reference
  m.state_dict() / m.load_state_dict(sd)       # the reference's 148 keys (weight_orig / weight_u / weight_v, running stats)
  m.train() / m.eval(), m.to(device), m.parameters() (names + tensors)
Training (what latent_conditioner.py:246-314 does around `latent_conditioner(x)`): `m.loss_backward(x, y1, y2)` runs the
forward, the loss 10*MSE(y1) + MSE(y2) and the backward and leaves `m.grads`; `modules.latent_conditioner` drives it."""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from .. import ops
from ..engine import SgvError

_GROUPS = (32, 16, 8, 4, 2, 1)


class _HalfRes:
    """Input gradient of a 1x1 stride-2 convolution kept at the resolution of its output: `half` [B, ceil(H/2), ceil(W/2), C] holds
    the values of the even pixels of the [B, H, W, C] gradient, every other pixel is zero.  conv1's input-gradient GEMM adds it
    in its epilogue (sgv_op_gemm_nt_add_s2); `full()` materialises it (sgv_op_col2im) for any other consumer."""

    def __init__(self, half, shape):
        self.half, self.shape = half, shape

    def full(self):
        return ops.col2im(self.half.reshape(-1, self.half.shape[-1]), self.shape, 1, 1, 2, 0)


def _num_groups(channels):
    for g in _GROUPS:
        if channels % g == 0 and g <= channels:
            return g
    return 1


class LatentConditionerImg:
    def __init__(self, latent_conditioner_filter, latent_dim_end, input_shape, latent_dim, size2,
                 latent_conditioner_data_shape, dropout_rate=0.3, use_attention=True, return_dict=False,
                 compute_dtype="bf16", seed=0, fused_params=True):
        self.filters = [int(v) for v in latent_conditioner_filter]
        self.latent_dim, self.size2, self.latent_dim_end = int(latent_dim), int(size2), int(latent_dim_end)
        self.dropout_rate, self.use_attention, self.return_dict = float(dropout_rate), bool(use_attention), bool(return_dict)
        self.compute_dtype = compute_dtype
        self.dt = ops.tdtype(compute_dtype)
        self.training = True
        # fused_params: spectral-norm power iteration, clipping and AdamW run as multi-tensor passes over fixed gradient
        # buffers (ops.ParamSet); otherwise one operator call per tensor (the mode the gradient parity tests read)
        self.fused_params = bool(fused_params)
        # implicit_conv: 3x3 / strided convolutions as implicit GEMMs (sgv_op_conv2d_nt); SGV_LC_IMPLICIT=0 keeps the
        # im2col + GEMM lowering (A/B runs, and the comparator of tests/test_ops_gpu.py)
        self.implicit_conv = os.environ.get("SGV_LC_IMPLICIT", "1") != "0"
        self.fused_tail = os.environ.get("SGV_LC_FUSED_TAIL", "1") != "0"
        self.pset = None
        self._conv_sums = None          # GroupNorm statistics left by the last _conv call (the direct stem produces them), else None
        if any(c % 16 for c in self.filters):
            raise SgvError("latent_conditioner_filter entries must be multiples of 16 (bottleneck channels feed 8-wide GEMM tiles)")
        self.blocks = []
        cin = self.filters[0]
        for i, cout in enumerate(self.filters[1:]):
            stride = 2 if i in (1, 3) else 1
            self.blocks.append(dict(i=i, cin=cin, cout=cout, mid=cout // 2, stride=stride,
                                    skip=(stride != 1 or cin != cout), se=self.use_attention and 2 <= i <= 4))
            cin = cout
        self.final_c = self.filters[-1]
        self.hidden = self.final_c * 2
        self.P = {}            # parameters and buffers, reference names, fp32 CUDA tensors
        self.grads = {}
        self._tape = None
        self._init_state(seed)

    # ---- parameters -------------------------------------------------------------------------------------------------
    def _spec(self):
        """(name, shape, kind) in the reference's state_dict order; kind: w (trainable), u/v (spectral-norm vectors), buf."""
        out = []

        def sn_conv(prefix, co, ci, k):
            out.extend([(prefix + ".weight_orig", (co, ci, k, k), "w"), (prefix + ".weight_u", (co,), "u"), (prefix + ".weight_v", (ci * k * k,), "v")])

        def gn(prefix, c):
            out.extend([(prefix + ".weight", (c,), "w"), (prefix + ".bias", (c,), "w")])

        def sn_lin(prefix, o, k):
            out.extend([(prefix + ".bias", (o,), "w"), (prefix + ".weight_orig", (o, k), "w"), (prefix + ".weight_u", (o,), "u"), (prefix + ".weight_v", (k,), "v")])

        def lin(prefix, o, k):
            out.extend([(prefix + ".weight", (o, k), "w"), (prefix + ".bias", (o,), "w")])

        def bn(prefix, c):
            out.extend([(prefix + ".weight", (c,), "w"), (prefix + ".bias", (c,), "w"), (prefix + ".running_mean", (c,), "buf"),
                        (prefix + ".running_var", (c,), "buf"), (prefix + ".num_batches_tracked", (), "buf")])
        sn_conv("initial_conv.0", self.filters[0], 1, 7)
        gn("initial_conv.1", self.filters[0])
        for b in self.blocks:
            p = f"layers.{b['i']}"
            sn_conv(p + ".conv1", b["mid"], b["cin"], 1)
            gn(p + ".gn1", b["mid"])
            sn_conv(p + ".conv2", b["cout"], b["mid"], 3)
            gn(p + ".gn2", b["cout"])
            if b["skip"]:
                sn_conv(p + ".skip.0", b["cout"], b["cin"], 1)
                gn(p + ".skip.1", b["cout"])
            if b["se"]:
                lin(p + ".se.fc1", b["cout"] // 16, b["cout"])
                lin(p + ".se.fc2", b["cout"], b["cout"] // 16)
        h = self.hidden
        sn_lin("feature_processor.1", h, self.final_c)
        gn("feature_processor.2", h)
        sn_lin("feature_processor.5", h, h)
        gn("feature_processor.6", h)
        for head, odim in (("latent_main", self.latent_dim_end), ("xs", self.latent_dim * self.size2)):
            l1, l2 = (head + "_layer1", head + "_layer2")
            sn_lin(l1 + ".0", h // 2, h)
            bn(l1 + ".1", h // 2)
            sn_lin(l2 + ".0", h // 4, h // 2)
            bn(l2 + ".1", h // 4)
            lin("main_skip_proj" if head == "latent_main" else "xs_skip_proj", h // 4, h)
            lin(head + "_output", odim, h // 4)
        return out

    def _init_state(self, seed):
        """Same distributions as the reference's `_init_weights` (kaiming_normal fan_out for convs and hidden Linears,
        xavier_normal for the two output layers, zero biases, unit norms); values come from numpy Philox, so parity
        tests load the reference's own state instead."""
        rng = np.random.Generator(np.random.Philox(seed))
        for name, shape, kind in self._spec():
            if kind == "u" or kind == "v":
                a = rng.standard_normal(shape).astype(np.float32)
                a /= max(float(np.linalg.norm(a)), 1e-12)
            elif name.endswith("num_batches_tracked"):
                a = np.zeros((), np.float32)
            elif name.endswith("running_var"):
                a = np.ones(shape, np.float32)
            elif name.endswith("running_mean") or name.endswith(".bias"):
                a = np.zeros(shape, np.float32)
            elif len(shape) == 1:
                a = np.ones(shape, np.float32)                      # GroupNorm / LayerNorm / BatchNorm weights
            elif len(shape) == 4:
                a = (rng.standard_normal(shape) * math.sqrt(2.0 / (shape[0] * shape[2] * shape[3]))).astype(np.float32)
            elif shape[0] in (self.latent_dim_end, self.latent_dim * self.size2):
                a = (rng.standard_normal(shape) * math.sqrt(2.0 / (shape[0] + shape[1]))).astype(np.float32)
            else:
                a = (rng.standard_normal(shape) * math.sqrt(2.0 / shape[0])).astype(np.float32)
            self.P[name] = torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def state_dict(self):
        out = {}
        for name, shape, kind in self._spec():
            t = self.P[name].detach().cpu().clone().reshape(tuple(shape))      # () for num_batches_tracked, as torch exports it
            out[name] = t.to(torch.int64) if name.endswith("num_batches_tracked") else t
        return out

    def load_state_dict(self, sd, strict=True):
        names = [n for n, _, _ in self._spec()]
        missing = [n for n in names if n not in sd]
        extra = [k for k in sd if k not in names]
        if strict and (missing or extra):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:4]}, unexpected {extra[:4]}")
        for name, shape, _ in self._spec():
            if name in sd:
                v = sd[name]
                a = torch.as_tensor(np.asarray(v.detach().cpu() if torch.is_tensor(v) else v), dtype=torch.float32)
                if tuple(a.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for {name}: {tuple(a.shape)} vs {tuple(shape)}")
                self.P[name] = a.reshape(max(1, a.numel())).contiguous().cuda() if a.dim() == 0 else a.contiguous().cuda()
        self.pset = None            # parameter tensors were replaced: rebuild the tables on next use
        return self

    def __getstate__(self):
        d = dict(self.__dict__)
        d["P"] = {k: v.detach().cpu().numpy() for k, v in self.P.items()}
        d["grads"], d["_tape"], d["pset"], d["_pending"], d["_conv_sums"] = {}, None, None, None, None
        d.pop("G", None)
        d.pop("_entry", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.P = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d["P"].items()}
        self.pset = None

    def named_parameters(self):
        return [(n, self.P[n]) for n, _, k in self._spec() if k == "w"]

    def parameters(self):
        return [t for _, t in self.named_parameters()]

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device=None, *a, **k):
        if device is not None and str(device).startswith("cpu"):
            raise SgvError("this model only runs on an MI355X: there is no CPU path")
        return self

    def apply(self, fn):
        """latent_conditioner.apply(safe_initialize_weights_He) (latent_conditioner.py:169-177,223) and
        latent_conditioner.apply(init_weights) (latent_conditioner_e2e.py:267-287).  On the spectrally normalised layers
        the reference's hook recomputes `.weight` from `weight_orig` at the next forward, so those calls only re-initialise
        the plain Linear layers (SE, skip projections, the two output layers): kaiming_uniform(relu) -- or, for
        `init_weights`, normal(0, 0.1) when out_features <= 64 -- and zero biases; norm layers get ones / zeros (their
        state at that point anyway; BatchNorm1d is not in `init_weights`' list).  Values come from numpy's generator (same
        distributions, different stream)."""
        e2e = getattr(fn, "__name__", "") == "init_weights"
        rng = np.random.default_rng(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        spec = {n: (sh, k) for n, sh, k in self._spec()}
        for n, (sh, k) in spec.items():
            if k != "w" or not n.endswith(".weight") or len(sh) != 2:
                continue                                    # only plain nn.Linear weights ([out, in], not weight_orig)
            if e2e and sh[0] <= 64:
                w = rng.normal(0.0, 0.1, sh)
            else:
                bound = math.sqrt(6.0 / sh[1])
                w = rng.uniform(-bound, bound, sh)
            self.P[n].copy_(torch.from_numpy(w.astype(np.float32)))
            bn_ = n[:-len("weight")] + "bias"
            if bn_ in self.P:
                self.P[bn_].zero_()
        return self

    # ---- fused parameter passes -----------------------------------------------------------------------------------
    def _build_pset(self):
        """One flat gradient arena (views per parameter) + the multi-tensor tables.  Normalised weights whose matrix has
        cols % 4 == 0 are registered as such (their arena slot then holds the gradient wrt W/sigma and the step applies the
        chain rule); the 7x7 stem (49 columns) keeps the per-tensor path."""
        names = [(n, sh) for n, sh, k in self._spec() if k == "w"]
        sizes = [(int(np.prod(sh)) + 3) // 4 * 4 for _, sh in names]
        self._arena = torch.zeros(sum(sizes), dtype=torch.float32, device="cuda")
        self.G, self._entry, entries, off = {}, {}, [], 0
        for (n, sh), sz in zip(names, sizes):
            numel = int(np.prod(sh))
            self.G[n] = self._arena[off:off + numel].view(sh)
            off += sz
            e = dict(p=self.P[n], g=self.G[n])
            if n.endswith("weight_orig"):
                rows, cols = sh[0], numel // sh[0]
                if cols % 4 == 0 and numel % 4 == 0:
                    pre = n[:-len(".weight_orig")]
                    e.update(rows=rows, cols=cols, u=self.P[pre + ".weight_u"], v=self.P[pre + ".weight_v"])
            if numel % 4:
                self.pset = False        # a tensor the vector kernels cannot take: stay on the per-tensor path
                return
            self._entry[n] = len(entries)
            entries.append(e)
        self.pset = ops.ParamSet(entries)

    def _fused(self):
        if not self.fused_params:
            return False
        if self.pset is None:
            self._build_pset()
        return bool(self.pset)

    def _is_fused_sn(self, prefix):
        return self._fused() and self.pset.entries[self._entry[prefix + ".weight_orig"]].get("rows", 0) > 0

    # ---- layer helpers: each returns (output, backward closure) ---------------------------------------------------
    def _acc(self, name, g):
        if self._fused():
            # fixed buffer the multi-tensor step reads: filed by ONE copy launch at the end of backward (_flush_grads)
            if getattr(self, "_pending", None) is None:
                self._pending = []
            self._pending.append((g.contiguous(), self.G[name]))
            self.grads[name] = self.G[name]
        else:
            self.grads[name] = g if name not in self.grads else ops.addf(self.grads[name], g)

    def _sn(self, prefix, Wm):
        """-> (sigma2 tensor or None, device address of 1/sigma)"""
        if self._is_fused_sn(prefix):          # sigma was computed for every layer at the start of this forward
            return None, self.pset.sigma_ptr(self._entry[prefix + ".weight_orig"]) + 4
        sig2 = ops.sn_power_iteration(Wm, self.P[prefix + ".weight_u"], self.P[prefix + ".weight_v"], self.training)
        return sig2, sig2.data_ptr() + 4

    def _conv(self, prefix, x4, k, stride, pad, need_dx=True):
        W = self.P[prefix + ".weight_orig"]
        co, ci = W.shape[0], W.shape[1]
        Wm = W.view(co, -1)
        sig2, inv_sigma = self._sn(prefix, Wm)
        Wp = ops.conv_weight_pack(W, self.dt)
        B, H, Wd, _ = x4.shape
        direct = (k == 1 and stride == 1 and ci % 8 == 0)
        # implicit GEMM (no im2col matrix): every convolution whose input channels fill 16-byte chunks, i.e. all but the stem
        implicit = self.implicit_conv and not direct and ci % 8 == 0 and k * k <= 24
        # the one-channel stem: direct convolution on the MFMA, which also leaves the GroupNorm statistics of its output
        stem = self.implicit_conv and ci == 1 and stride == 1 and k <= 7 and 2 * pad == k - 1 and self.dt == torch.bfloat16 and not need_dx
        col = None
        self._conv_sums = None
        if direct:
            col, Ho, Wo = x4.view(-1, ci), H, Wd
        if stem:
            y, self._conv_sums = ops.stem_conv(x4.view(B, H, Wd), Wp, co, k, pad, _num_groups(co), scale=inv_sigma)
            Ho, Wo = H, Wd
        elif implicit:
            y = ops.conv2d_nt(x4, Wp, co, k, k, stride, pad, k * k * ci, ci, scale=inv_sigma)
            Ho, Wo = y.shape[1], y.shape[2]
        else:
            if col is None:
                col, Ho, Wo = ops.im2col(x4, k, k, stride, pad)
            y = ops.gemm_nt(col, Wp, scale=inv_sigma).view(B, Ho, Wo, co)
        u, v = self.P[prefix + ".weight_u"], self.P[prefix + ".weight_v"]

        def bwd(dy4, addend=None):
            """-> dX (+ addend, a tensor of the input's shape added in the GEMM epilogue of the 1x1 stride-1 case)"""
            dy = dy4.reshape(-1, co)
            if implicit:
                G = ops.conv_weight_unpack(ops.conv2d_tn(dy4.reshape(B, Ho, Wo, co), x4, k, k, stride, pad), W.shape)
            elif stem:
                G = ops.conv_weight_unpack(ops.stem_conv_dw(x4.view(B, H, Wd), dy4.reshape(B, Ho, Wo, co), k, pad), W.shape)
            else:
                G = ops.conv_weight_unpack(ops.gemm_tn(dy, col), W.shape)       # gradient wrt W / sigma
            if sig2 is None:
                self._acc(prefix + ".weight_orig", G)                            # chain rule applied by the fused step
            else:
                self._acc(prefix + ".weight_orig", ops.sn_grad(G.view(co, -1), u, v, Wm, sig2).view(W.shape))
            if not need_dx:
                return None
            Wt = ops.transpose(Wp.view(1, co, -1), self.dt, 1, co, Wp.shape[1]).view(Wp.shape[1], co)
            if isinstance(addend, _HalfRes) and not (direct and self.dt == torch.bfloat16 and (ci < 256 or co < 2048)):
                addend = addend.full()          # no epilogue for it here: materialise the zero-filled full-resolution tensor
            if implicit and stride == 1 and co % 8 == 0:
                # dX of a stride-1 convolution = convolution of dY with the reversed taps of the transposed weights
                dx = ops.conv2d_nt(dy4.reshape(B, Ho, Wo, co), Wt, ci, k, k, 1, k - 1 - pad, co, ci * co, flip=True, scale=inv_sigma)
            elif direct and isinstance(addend, _HalfRes):
                # + the stride-2 skip projection's input gradient, read at half resolution by the GEMM epilogue
                return ops.gemm_nt_add_s2(dy, Wt, addend.half, H, Wd, scale=inv_sigma).view(x4.shape)
            elif direct:
                return ops.gemm_nt(dy, Wt, scale=inv_sigma, addend=None if addend is None else addend.view(-1, ci)).view(x4.shape)
            elif implicit and k == 1 and stride == 2 and pad == 0 and addend is None and self.fused_tail:
                # 1x1 stride-2 projection: its input gradient is non-zero at the even pixels only; hand the compact tensor on
                return _HalfRes(ops.gemm_nt(dy, Wt, scale=inv_sigma).view(B, Ho, Wo, ci), tuple(x4.shape))
            else:
                dx = ops.col2im(ops.gemm_nt(dy, Wt, scale=inv_sigma), x4.shape, k, k, stride, pad)
            return dx if addend is None else ops.add(dx, addend)
        return y, bwd

    def _gn(self, prefix, y4, act, sums=None):
        """sums: statistics the producing kernel already computed (the stem convolution)"""
        B, H, Wd, Cc = y4.shape
        G = _num_groups(Cc)
        gamma, beta = self.P[prefix + ".weight"], self.P[prefix + ".bias"]
        y3 = y4.view(B, H * Wd, Cc)
        if sums is not None:
            out = ops.gn_apply(y3, G, gamma, beta, sums, act)
        else:
            out, sums = ops.gn_fwd(y3, G, gamma, beta, act)
        return out.view(y4.shape), self._gn_bwd(prefix, y4, sums, act)

    def _gn_bwd(self, prefix, y4, sums, act):
        """backward closure of out = act(gn(y4)) given the forward statistics"""
        B, H, Wd, Cc = y4.shape
        G = _num_groups(Cc)
        gamma, beta = self.P[prefix + ".weight"], self.P[prefix + ".bias"]
        y3 = y4.view(B, H * Wd, Cc)

        def bwd(dout4):
            dg, db = torch.empty_like(gamma), torch.empty_like(beta)          # written, not accumulated: no zero-fill launches
            dy = ops.gn_bwd(y3, dout4.reshape(B, H * Wd, Cc), G, gamma, beta, sums, act, dg, db, accumulate=False)
            self._acc(prefix + ".weight", dg)
            self._acc(prefix + ".bias", db)
            return dy.view(y4.shape)
        return bwd

    def _linear(self, prefix, x, sn, act=ops.LIN_NONE):
        W = self.P[prefix + (".weight_orig" if sn else ".weight")]
        b = self.P[prefix + ".bias"]
        sig2, scale = self._sn(prefix, W) if sn else (None, None)
        y = ops.linear_fwd(x, W, b, scale, act)

        def bwd(dy):
            dz = ops.act_bwd(y, dy, act) if act != ops.LIN_NONE else dy
            if sn:       # sn_grad wants G = dz^T x, the gradient wrt W / sigma: one weight-gradient pass (unscaled) + the scaled dX
                dx = ops.linear_bwd(dz, x, W, scale, need_dw=False)[0]
                _, G, db = ops.linear_bwd(dz, x, W, None, need_dx=False)
            else:
                dx, dW, db = ops.linear_bwd(dz, x, W, scale)    # dx = scale * dz W ; dW = scale * dz^T x
            self._acc(prefix + ".bias", db)
            if sn:
                if sig2 is None:
                    self._acc(prefix + ".weight_orig", G)
                else:
                    self._acc(prefix + ".weight_orig", ops.sn_grad(G, self.P[prefix + ".weight_u"], self.P[prefix + ".weight_v"], W, sig2))
            else:
                self._acc(prefix + ".weight", dW)
            return dx
        return y, bwd

    def _dropout(self, x, p, masks):
        if not self.training or p == 0.0:
            return x, (lambda d: d)
        if masks is not None:
            mask = masks.pop(0)
            if tuple(mask.shape) != tuple(x.shape):
                raise ValueError(f"dropout mask of shape {tuple(mask.shape)} for an activation of shape {tuple(x.shape)}")
        else:
            mask = (torch.rand(x.shape, device=x.device) >= p).float()
        scale = 1.0 / (1.0 - p)
        return ops.mask_scale(x, mask, scale), (lambda d: ops.mask_scale(d, mask, scale))

    def _layernorm(self, prefix, x):
        gamma, beta = self.P[prefix + ".weight"], self.P[prefix + ".bias"]
        y, stat = ops.layernorm_fwd(x, gamma, beta)

        def bwd(dy):
            dx, dg, db = ops.layernorm_bwd(x, gamma, stat, dy)
            self._acc(prefix + ".weight", dg)
            self._acc(prefix + ".bias", db)
            return dx
        return y, bwd

    def _batchnorm(self, prefix, x):
        gamma, beta = self.P[prefix + ".weight"], self.P[prefix + ".bias"]
        train = self.training
        y, stat = ops.batchnorm_fwd(x, gamma, beta, self.P[prefix + ".running_mean"], self.P[prefix + ".running_var"], train)
        if train:
            self.P[prefix + ".num_batches_tracked"] = self.P[prefix + ".num_batches_tracked"] + 1

        def bwd(dy):
            dx, dg, db = ops.batchnorm_bwd(x, gamma, stat, dy, train)
            self._acc(prefix + ".weight", dg)
            self._acc(prefix + ".bias", db)
            return dx
        return y, bwd

    def _relu(self, x):
        y = ops.act_fwd(x, ops.LIN_RELU)
        return y, (lambda d: ops.act_bwd(y, d, ops.LIN_RELU))

    def _block(self, b, x4):
        p = f"layers.{b['i']}"
        c1, bw_c1 = self._conv(p + ".conv1", x4, 1, 1, 0)
        a1, bw_g1 = self._gn(p + ".gn1", c1, ops.ACT_RELU_GN)
        c2, bw_c2 = self._conv(p + ".conv2", a1, 3, b["stride"], 1)
        B, H, Wd, Cc = c2.shape
        # fused_tail: the block's tail -- normalise the skip projection (and, without squeeze-excite, the main branch), scale,
        # add, relu -- is one pass over the rows (sgv_op_gn_tail) instead of three or four
        tail = self.fused_tail and b["skip"]
        if b["se"] or not tail:
            o2, bw_g2 = self._gn(p + ".gn2", c2, ops.ACT_NONE)
        if b["se"]:
            o2f = o2.view(B, H * Wd, Cc)
            pooled = ops.avgpool_fwd(o2f)
            hid, bw_f1 = self._linear(p + ".se.fc1", pooled, False, ops.LIN_RELU)
            s, bw_f2 = self._linear(p + ".se.fc2", hid, False, ops.LIN_SIGMOID)
        if b["skip"]:
            sc, bw_sc = self._conv(p + ".skip.0", x4, 1, b["stride"], 0)
        if tail:
            G = _num_groups(Cc)
            sc3 = sc.view(B, H * Wd, Cc)
            g_s, b_s = self.P[p + ".skip.1.weight"], self.P[p + ".skip.1.bias"]
            if b["se"]:
                out, sums_s, _ = ops.gn_tail(sc3, G, g_s, b_s, o2f, cscale=s)
            else:
                out, sums_s, sums_2 = ops.gn_tail(sc3, G, g_s, b_s, c2.view(B, H * Wd, Cc), self.P[p + ".gn2.weight"], self.P[p + ".gn2.bias"])
                bw_g2 = self._gn_bwd(p + ".gn2", c2, sums_2, ops.ACT_NONE)
            bw_sg = self._gn_bwd(p + ".skip.1", sc, sums_s, ops.ACT_NONE)
            out = out.view(c2.shape)
        else:
            o3 = ops.chan_scale_fwd(o2f, s).view(o2.shape) if b["se"] else o2
            if b["skip"]:
                sk, bw_sg = self._gn(p + ".skip.1", sc, ops.ACT_NONE)
            else:
                sk = x4
            out = ops.add_relu(o3, sk)

        def bwd(dout):
            d = ops.relu_bwd(out, dout)
            if b["se"]:
                dx_scale, ds = ops.chan_scale_bwd(o2f, s, d.view(B, H * Wd, Cc))
                dpool = bw_f1(bw_f2(ds))
                d_o2 = ops.avgpool_bwd(dpool, dx_scale).view(c2.shape)
            else:
                d_o2 = d
            dx_skip = bw_sc(bw_sg(d)) if b["skip"] else d
            return bw_c1(bw_g1(bw_c2(bw_g2(d_o2))), addend=dx_skip)          # + dx_skip in the epilogue of conv1's dX GEMM
        return out, bwd

    # ---- forward / backward -----------------------------------------------------------------------------------------
    def forward(self, x, dropout_masks=None):
        """x: [B, H*W] (or anything reshapeable to it) -> (latent_main [B, latent_dim_end], xs [B, size2, latent_dim])."""
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        x = x.to(device="cuda", dtype=torch.float32)
        B = x.shape[0]
        side = int(math.sqrt(x.shape[-1]))
        x = x.reshape(B, side, side)
        if float(x.min()) < -0.1:          # reference forward: inputs in [-1, 1] are mapped to [0, 1]
            x = (x + 1) / 2
        masks = list(dropout_masks) if dropout_masks is not None else None
        self.grads = {}
        if self._fused():
            self.pset.power_iteration(self.training)       # every normalised layer at once (legacy hook semantics per module)
        back = []
        x4 = x.to(self.dt).contiguous().view(B, side, side, 1)
        c0, bw = self._conv("initial_conv.0", x4, 7, 1, 3, need_dx=False)
        back.append(bw)
        if self._conv_sums is not None and self.fused_tail:
            # GroupNorm + ReLU + MaxPool in one pass over the convolution output (the normalised activation is never stored:
            # the backward needs the arg-max, the convolution output and its statistics only)
            h, pool_idx = ops.gn_relu_maxpool_fwd(c0, _num_groups(c0.shape[-1]), self.P["initial_conv.1.weight"], self.P["initial_conv.1.bias"],
                                                  self._conv_sums)
            back.append(self._gn_bwd("initial_conv.1", c0, self._conv_sums, ops.ACT_RELU_GN))
        else:
            a0, bw = self._gn("initial_conv.1", c0, ops.ACT_RELU_GN, sums=self._conv_sums)
            back.append(bw)
            h, pool_idx = ops.maxpool_fwd(a0)
        back.append(lambda d, idx=pool_idx, shp=tuple(c0.shape): ops.maxpool_bwd(idx, d, shp))
        for b in self.blocks:
            h, bw = self._block(b, h)
            back.append(bw)
        Bh, H, Wd, Cc = h.shape
        feat = ops.avgpool_fwd(h.view(Bh, H * Wd, Cc))
        back.append(lambda d, shp=(Bh, H * Wd, Cc), s4=h.shape: ops.avgpool_bwd(d, shp, self.dt).view(s4))
        r = self.dropout_rate
        f = feat
        for step in (lambda t: self._dropout(t, r * 0.3, masks), lambda t: self._linear("feature_processor.1", t, True),
                     lambda t: self._layernorm("feature_processor.2", t), self._relu, lambda t: self._dropout(t, r * 0.4, masks),
                     lambda t: self._linear("feature_processor.5", t, True), lambda t: self._layernorm("feature_processor.6", t),
                     self._relu, lambda t: self._dropout(t, r * 0.4, masks)):
            f, bw = step(f)
            back.append(bw)
        features = f

        def head(name, skip_name, out_name):
            t, chain = features, []
            for step in (lambda z: self._linear(name + "_layer1.0", z, True), lambda z: self._batchnorm(name + "_layer1.1", z), self._relu,
                         lambda z: self._dropout(z, r * 0.3, masks),
                         lambda z: self._linear(name + "_layer2.0", z, True), lambda z: self._batchnorm(name + "_layer2.1", z), self._relu,
                         lambda z: self._dropout(z, 0.2, masks)):
                t, bw_ = step(t)
                chain.append(bw_)
            sk, bw_sk = self._linear(skip_name, features, False)
            comb = ops.addf(t, sk)
            o, bw_o = self._linear(out_name, comb, False)

            def bwd(do):
                dc = bw_o(do)
                d = dc
                for fn in reversed(chain):
                    d = fn(d)
                return ops.addf(d, bw_sk(dc))
            return o, bwd
        main, bw_main = head("latent_main", "main_skip_proj", "latent_main_output")
        xs, bw_xs = head("xs", "xs_skip_proj", "xs_output")

        def backward(d_main, d_xs):
            d = ops.addf(bw_main(d_main.contiguous()), bw_xs(d_xs.reshape(B, -1).contiguous()))
            for fn in reversed(back):
                d = fn(d)
                if d is None:
                    break
            return self.grads
        self._tape = backward
        xs3 = xs.view(B, self.size2, self.latent_dim)
        if self.return_dict:
            return {"latent_main": main, "xs": xs3, "features": features}
        return main, xs3

    __call__ = forward

    def backward(self, d_main, d_xs):
        if self._tape is None:
            raise SgvError("backward() needs a preceding forward()")
        g = self._tape(d_main, d_xs)
        self._tape = None
        self._flush_grads()
        return g

    def _flush_grads(self):
        if getattr(self, "_pending", None):
            ops.multi_copy(self._pending)
            self._pending = []

    def loss_backward(self, x, y1, y2, dropout_masks=None, w1=10.0, w2=1.0, preds=None):
        """latent_conditioner.py:285-301: forward, A = MSE(y_pred1, y1), B = MSE(y_pred2, y2), loss = w1*A + w2*B
        (10 and 1 there; the end-to-end loop uses reg_weight*0.9 and reg_weight*0.1), backward.  `preds` = the outputs of a
        forward already run on this input in training mode (its tape is still current).
        Returns (loss, A, B) as floats; gradients are left in `self.grads` keyed by parameter name."""
        p1, p2 = preds if preds is not None else self.forward(x, dropout_masks)
        y1 = torch.as_tensor(y1).to(device="cuda", dtype=torch.float32).contiguous()
        y2 = torch.as_tensor(y2).to(device="cuda", dtype=torch.float32).contiguous()
        la, d1 = ops.mse(p1, y1, gscale=float(w1))
        lb, d2 = ops.mse(p2.reshape(p2.shape[0], -1), y2.reshape(y2.shape[0], -1), gscale=float(w2))
        self.backward(d1, d2)
        A, Bv = float(la), float(lb)
        return float(w1) * A + float(w2) * Bv, A, Bv
