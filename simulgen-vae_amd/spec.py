"""Model configuration and the ordered parameter/buffer specification.

The key names, shapes and ordering reproduce the `state_dict()` of the reference
`modules.VAE_network.VAE` after `model.apply(add_sn)` (reference
modules/VAE_network.py:60-77, modules/encoder.py:119-144, modules/decoder.py:105-168,
modules/common.py:15-37,78-162), so a checkpoint written by either side loads in the other.
Nothing here depends on torch.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

LOSS_IDS = {"MSE": 0, "MAE": 1, "smoothL1": 2, "Huber": 3}


def gn_groups(c: int) -> int:
    """GroupNorm group count used everywhere in the reference: min(8, max(1, C//4))."""
    return min(8, max(1, c // 4))


@dataclass
class VAEConfig:
    latent_dim: int
    hierarchical_dim: int
    num_filter_enc: List[int]
    num_filter_dec: List[int]
    num_node: int
    num_time: int
    lossfun: str = "MSE"
    small: bool = True

    def __post_init__(self):
        self.num_filter_enc = [int(v) for v in self.num_filter_enc]
        self.num_filter_dec = [int(v) for v in self.num_filter_dec]
        if self.lossfun not in LOSS_IDS:
            # reference falls back to MSE for unknown names (VAE_network.py:110)
            self.lossfun = "MSE"


@dataclass
class Entry:
    name: str
    shape: Tuple[int, ...]
    kind: str  # bias | weight_orig | weight_u | weight_v | gn_weight | gn_bias
    layer: str  # owning layer prefix
    trainable: bool


@dataclass
class LayerDesc:
    """One Conv1d / ConvTranspose1d / Linear / GroupNorm of the reference graph."""
    prefix: str
    op: str  # conv | convT | linear | gn
    cin: int = 0
    cout: int = 0
    k: int = 1
    groups: int = 0
    used_in_forward: bool = True   # spectral-norm power iteration runs (module is called)
    has_grad: bool = True          # receives a gradient in training


def _conv(prefix, cin, cout, k, transposed=False, **kw):
    return LayerDesc(prefix, "convT" if transposed else "conv", cin, cout, k, **kw)


def _lin(prefix, fin, fout, **kw):
    return LayerDesc(prefix, "linear", fin, fout, 1, **kw)


def _gn(prefix, c, **kw):
    return LayerDesc(prefix, "gn", c, c, 1, gn_groups(c), **kw)


def layer_list(cfg: VAEConfig) -> List[LayerDesc]:
    """All parameterised layers in reference registration order."""
    enc, dec = cfg.num_filter_enc, cfg.num_filter_dec
    T, H, Z = cfg.num_time, cfg.hierarchical_dim, cfg.latent_dim
    L: List[LayerDesc] = []
    n_enc = len(enc)
    # --- Encoder (encoder.py:119-144) ---
    for i in range(n_enc):
        cin = cfg.num_node if i == 0 else enc[i - 1]
        p = f"encoder.encoder_blocks.{i}.module_list.0._seq"
        L += [_conv(f"{p}.0", cin, enc[i], 1), _gn(f"{p}.1", enc[i])]
        if not cfg.small:
            L += [_conv(f"{p}.3", enc[i], enc[i], 3), _gn(f"{p}.4", enc[i])]
    for i in range(n_enc):
        p = f"encoder.encoder_residual_blocks.{i}.seq"
        L += [_conv(f"{p}.0", enc[i], enc[i], 3), _gn(f"{p}.1", enc[i])]
        if not cfg.small:
            L += [_conv(f"{p}.3", enc[i], enc[i], 3), _gn(f"{p}.4", enc[i])]
    for i in range(n_enc):
        # xs_linear.{n-1} output is dropped (encoder.py:167); xs_linear.0 feeds xs[2] which the
        # decoder never reads (decoder.py:184-185): both are called but get grad None.
        dead = (i == 0) or (i == n_enc - 1)
        L.append(_lin(f"encoder.xs_linear.{i}", enc[i] * T, H, has_grad=not dead))
    L.append(_lin("encoder.last_x_linear", enc[-1] * T, 2 * Z))
    # --- Decoder (decoder.py:105-168) ---
    n_st = len(dec) - 1
    for i in range(n_st):
        L.append(_conv(f"decoder.decoder_blocks.{i}.module_list.0._seq.0", dec[i], dec[i + 1], 3,
                       transposed=True))
    for i in range(n_st):
        c = dec[i + 1]
        p = f"decoder.decoder_residual_blocks.{i}.seq"
        if cfg.small:
            L += [_conv(f"{p}.0", c, 5 * c, 1), _gn(f"{p}.1", 5 * c),
                  _conv(f"{p}.3", 5 * c, 5 * c, 5), _gn(f"{p}.4", 5 * c),
                  _conv(f"{p}.6", 5 * c, c, 1), _gn(f"{p}.7", c)]
        else:
            L += [_conv(f"{p}.0", c, c, 1), _gn(f"{p}.1", c),
                  _conv(f"{p}.3", c, 5 * c, 5), _gn(f"{p}.4", 5 * c),
                  _conv(f"{p}.6", 5 * c, 5 * c, 5), _gn(f"{p}.7", 5 * c),
                  _conv(f"{p}.9", 5 * c, c, 1), _gn(f"{p}.10", c)]
    L += [_conv("decoder.recon.0", dec[-1], cfg.num_node, 1), _gn("decoder.recon.1", cfg.num_node)]
    L += [_lin("decoder.sequence_start.0.0", Z, Z * T),
          _conv("decoder.sequence_start.0.2", Z, dec[0], 5), _gn("decoder.sequence_start.0.3", dec[0])]
    for i in range(n_st):
        live = i < n_st - 1  # last stage breaks before conditioning (decoder.py:184-185)
        kw = dict(used_in_forward=live, has_grad=live)
        p = f"decoder.xs_sequence.{i}"
        L += [_lin(f"{p}.0", H, H * T, **kw), _conv(f"{p}.2", H, dec[i + 1], 5, **kw),
              _gn(f"{p}.3", dec[i + 1], **kw)]
    for name, mult in (("condition_z", 1), ("condition_xz", 2)):
        for i in range(n_st):
            live = i < n_st - 1
            kw = dict(used_in_forward=live, has_grad=live)
            c = mult * dec[i + 1]
            p = f"decoder.{name}.{i}"
            L += [_conv(f"{p}.0._seq.0", c, c, 3, **kw), _gn(f"{p}.0._seq.1", c, **kw)]
            if not cfg.small:
                L += [_conv(f"{p}.0._seq.3", c, c, 3, **kw), _gn(f"{p}.0._seq.4", c, **kw)]
            L.append(_conv(f"{p}.2", c, 2 * dec[i + 1], 3, **kw))
    return L


def sn_matrix_shape(l: LayerDesc) -> Tuple[int, int]:
    """(rows, cols) of the matrix legacy spectral_norm sees (torch nn/utils/spectral_norm.py:
    reshape_weight_to_matrix; dim=1 for ConvTranspose1d, 0 otherwise)."""
    return (l.cout, l.cin * l.k)


def weight_shape(l: LayerDesc) -> Tuple[int, ...]:
    if l.op == "conv":
        return (l.cout, l.cin, l.k)
    if l.op == "convT":
        return (l.cin, l.cout, l.k)
    if l.op == "linear":
        return (l.cout, l.cin)
    raise ValueError(l.op)


def param_spec(cfg: VAEConfig) -> List[Entry]:
    """state_dict entries in reference order: per module bias, weight_orig (parameters), then
    weight_u, weight_v (buffers); GroupNorm weight, bias."""
    out: List[Entry] = []
    for l in layer_list(cfg):
        if l.op == "gn":
            out.append(Entry(f"{l.prefix}.weight", (l.cout,), "gn_weight", l.prefix, l.has_grad))
            out.append(Entry(f"{l.prefix}.bias", (l.cout,), "gn_bias", l.prefix, l.has_grad))
        else:
            r, c = sn_matrix_shape(l)
            out.append(Entry(f"{l.prefix}.bias", (l.cout,), "bias", l.prefix, l.has_grad))
            out.append(Entry(f"{l.prefix}.weight_orig", weight_shape(l), "weight_orig", l.prefix, l.has_grad))
            out.append(Entry(f"{l.prefix}.weight_u", (r,), "weight_u", l.prefix, False))
            out.append(Entry(f"{l.prefix}.weight_v", (c,), "weight_v", l.prefix, False))
    return out


def num_params(cfg: VAEConfig) -> int:
    n = 0
    for e in param_spec(cfg):
        if e.kind in ("bias", "weight_orig", "gn_weight", "gn_bias"):
            k = 1
            for s in e.shape:
                k *= s
            n += k
    return n
