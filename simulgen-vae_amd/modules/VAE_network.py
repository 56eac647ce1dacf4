"""`VAE` with the constructor / method surface of the reference's modules.VAE_network.VAE
(modules/VAE_network.py:33-164), backed by the MI355X engine (libsgvae.so).

  VAE(latent_dim, hierarchical_dim, num_filter_enc, num_filter_dec, num_node, num_time,
      lossfun='MSE', batch_size=1, small=False, use_checkpointing=False)
  .forward(x) / __call__(x) -> (x_hat, recon_loss, [kl, kl2_0, ...], recon_loss_MSE)
  .encoder(x) -> (mu, log_var, xs) ; .decoder(z, xs, mode="random"|"fix", freeze_level=-1) -> (x_hat, kl_list)
  .state_dict() / .load_state_dict() with the reference's key names, .train()/.eval(), .to(),
  .compile_model(mode) (no-op: there is no tracing compiler here), picklable.
Differences a caller can observe: tensors returned are detached (gradients live inside the engine and
are applied by `modules.train.train` / `VAE.training_step`), and the model only runs on a GPU.
"""
from __future__ import annotations

import numpy as np
import torch

from ..engine import Engine, SgvError
from ..init import init_state
from ..spec import VAEConfig


class VAE:
    def __init__(self, latent_dim, hierarchical_dim, num_filter_enc, num_filter_dec, num_node, num_time,
                 lossfun="MSE", batch_size=1, small=False, use_checkpointing=False, compute_dtype="bf16", seed=7):
        self.cfg = VAEConfig(latent_dim, hierarchical_dim, list(num_filter_enc), list(num_filter_dec), num_node,
                             num_time, lossfun, bool(small))
        self.latent_dim = latent_dim
        self.lossfun = self.cfg.lossfun
        # The reference accepts the flag and forces it to False (VAE_network.py:60,68: no recompute exists in its code), so a caller
        # that passes True runs there without recompute.  Same here: accepted, forced to False, with a warning that says where
        # the memory went instead (every activation of a step stays resident: large model, batch 16, full size = 14.3 GiB of the
        # 288 GB of one MI355X)
        if use_checkpointing:
            import warnings
            warnings.warn("use_checkpointing=True is accepted and ignored, as in the reference (modules/VAE_network.py:68 forces it to "
                          "False): the engine keeps every activation of a step resident in HBM; Engine.memory_info() reports the bytes",
                          stacklevel=2)
        self.use_checkpointing = False
        self.batch_size = int(batch_size)
        self.compute_dtype = compute_dtype
        self.training = True
        self._engine = None
        self._pending_state = init_state(self.cfg, seed, reference_init=True)   # He-uniform weights, zero biases
        self.encoder = self._encode
        self.decoder = self._decode

    # ---- engine lifecycle ----
    def _eng(self, batch=None) -> Engine:
        need = max(self.batch_size, batch or 1)
        if self._engine is not None and need > self._engine.max_batch:
            self._pending_state = self._engine.state_dict()
            self._engine.close()
            self._engine = None
        if self._engine is None:
            self._engine = Engine(self.cfg, max_batch=need, compute_dtype=self.compute_dtype)
            self._engine.load_state(self._pending_state)
            self._pending_state = None
            self.batch_size = need
        return self._engine

    def to(self, *args, **kwargs):
        device = args[0] if args else kwargs.get("device", None)
        if device is not None and str(device).startswith("cpu"):
            raise SgvError("this VAE only runs on an MI355X: there is no CPU path")
        return self

    def cuda(self):
        return self

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def compile_model(self, mode="max-autotune"):
        return None

    def apply(self, fn):
        """model.apply(initialize_weights_He) / model.apply(add_sn) (train.py:71-72): the engine is always
        He-initialised and spectrally normalised, so these are accepted as no-ops."""
        return self

    def parameters(self):
        """nn.Module.parameters(): the trainable tensors (weight_orig / weight / bias entries of the state_dict, in its order) as
        detached copies -- enough for the reference's uses that only count or inspect them (torchinfo-style summaries,
        `sum(p.numel() for p in model.parameters())`); gradients and the optimizer live inside the engine."""
        sd = self.state_dict()
        return iter([v for k, v in sd.items() if not (k.endswith("weight_u") or k.endswith("weight_v"))])

    # ---- state ----
    def state_dict(self):
        sd = self._engine.state_dict() if self._engine is not None else self._pending_state
        return {k: torch.from_numpy(np.array(v, copy=True)) for k, v in sd.items()}

    def load_state_dict(self, sd, strict=True):
        st = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in sd.items()}
        if self._engine is not None:
            self._engine.load_state(st)
        else:
            self._pending_state = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in st.items()}
        return self

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_pending_state"] = {k: v.numpy() for k, v in self.state_dict().items()}
        d["_engine"] = None
        d.pop("encoder", None)
        d.pop("decoder", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self.encoder = self._encode
        self.decoder = self._decode

    # ---- forward paths ----
    @staticmethod
    def _prep(x):
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        return x.to(device="cuda", dtype=torch.float32).contiguous()

    def forward(self, x):
        try:
            x = self._prep(x)
            eng = self._eng(x.shape[0])
            eng.set_input(x)
            sc = eng.forward(train=self.training)
            x_hat = eng.xhat()
            dev = x.device
            kls = [torch.tensor(k, device=dev) for k in sc["kls"]]
            return x_hat, torch.tensor(sc["recon"], device=dev), kls, torch.tensor(sc["mse"], device=dev)
        except RuntimeError as e:  # same reporting as VAE_network.py:119-121
            print(f"Error in VAE forward pass: {e}")
            raise

    __call__ = forward

    def _encode(self, x):
        x = self._prep(x)
        eng = self._eng(x.shape[0])
        eng.set_input(x)
        mu, lv, xs = eng.encode()
        t = lambda a: torch.from_numpy(a).to(x.device)
        return t(mu), t(lv), [t(v) for v in xs]

    def _decode(self, z, xs=None, mode="random", freeze_level=-1):
        """Decoder.forward(z, xs, mode, freeze_level) (modules/decoder.py:170-216).  Two corners of that signature are refused
        rather than silently computed differently:
          * xs=None: the reference then never replaces z, so stage 1 adds the [B, latent] vector to a [B, C, T] map
            (decoder.py:179) -- a broadcasting RuntimeError for every size the path is used at (latent 32, T 200); no caller
            does it.  RuntimeError here too.
          * freeze_level >= 1 with mode="fix": the reference re-uses latents it stored on the module in EARLIER calls
            (decoder.py:202-207, `self.zs`, read at index i+1) -- state that survives across calls and batches; no caller passes
            it (utils.py:499, latent_conditioner_e2e.py:371, reconstruction_evaluator.py:174 all use the default -1).
            NotImplementedError instead of ignoring the argument.  freeze_level <= 0 is the plain path in the reference as well
            (`i < freeze_level` is never true)."""
        if xs is None:
            raise RuntimeError("decoder(z, xs=None): the reference adds the [B, latent] vector to a [B, C, T] map in this case "
                               "(modules/decoder.py:179) and fails; pass the encoder's xs")
        if mode == "fix" and freeze_level is not None and freeze_level >= 1:
            raise NotImplementedError("decoder(..., mode='fix', freeze_level >= 1): the reference's cross-call latent cache "
                                      "(modules/decoder.py:202-207) is not reproduced; use the default freeze_level=-1")
        eng = self._eng(z.shape[0])
        kls = eng.decode(self._prep(z), [self._prep(v) for v in xs], fix=(mode == "fix"))
        return eng.xhat(), [torch.tensor(k, device="cuda") for k in kls]

    # ---- one optimisation step (what train.py:139-168 does around model(image)) ----
    def training_step(self, x, alpha, beta, lr, allreduce=None, sync=True):
        x = self._prep(x)
        eng = self._eng(x.shape[0])
        eng.set_input(x)
        sc = eng.forward(train=True, sync=sync)
        if allreduce is not None:
            eng.backward(alpha, beta)
            allreduce.step(eng, lr)
        else:
            eng.backward_step(alpha, beta, lr)
        return sc
