"""Finite-difference check of the latent conditioner's block backward (debug aid).
   python tests/micro/lc_fd_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import simulgen_vae_amd
from simulgen_vae_amd.modules.latent_conditioner_model_cnn import LatentConditionerImg
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "lc_small.npz"))
m = LatentConditionerImg([int(v) for v in g["filters"]], 32, (1, 32, 32), 8, 3, (32, 32), compute_dtype="f32")
m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("s0.")})
m.eval()          # no power iteration: the function is fixed
torch.manual_seed(0)
for b in m.blocks:
    side = {0: 16, 1: 16, 2: 8, 3: 8, 4: 4}[b["i"]]
    x = torch.randn(4, side, side, b["cin"], device="cuda")
    D = torch.randn_like(x)
    out, bwd = m._block(b, x)
    R = torch.randn_like(out)
    m.grads = {}
    dx = bwd(R)
    an = float((dx.double() * D.double()).sum())
    eps = 1e-2
    lp = float((m._block(b, x + eps * D)[0].double() * R.double()).sum())
    lm = float((m._block(b, x - eps * D)[0].double() * R.double()).sum())
    fd = (lp - lm) / (2 * eps)
    print(f"block {b['i']} stride {b['stride']} se {b['se']}: analytic {an:.5f} finite-diff {fd:.5f} rel {abs(an-fd)/abs(fd):.2e}")
# stem
x = torch.rand(4, 32 * 32, device="cuda")


def fd(fn, x, name, eps=1e-2):
    D = torch.randn_like(x)
    out, bwd = fn(x)
    R = torch.randn_like(out)
    m.grads = {}
    dx = bwd(R)
    an = float((dx.double() * D.double()).sum())
    lp = float((fn(x + eps * D)[0].double() * R.double()).sum())
    lm = float((fn(x - eps * D)[0].double() * R.double()).sum())
    f = (lp - lm) / (2 * eps)
    print(f"{name}: analytic {an:.5f} fd {f:.5f} rel {abs(an - f) / abs(f):.2e}")


x8 = torch.randn(4, 8, 8, 64, device="cuda")
fd(lambda t: m._conv("layers.3.conv1", t, 1, 1, 0), x8, "conv 1x1 s1 (64->32)")
fd(lambda t: m._conv("layers.3.skip.0", t, 1, 2, 0), x8, "conv 1x1 s2 (64->64)")
x8b = torch.randn(4, 8, 8, 32, device="cuda")
fd(lambda t: m._conv("layers.3.conv2", t, 3, 2, 1), x8b, "conv 3x3 s2 (32->64)")
fd(lambda t: m._conv("layers.2.conv2", t, 3, 1, 1), x8b, "conv 3x3 s1 (32->64)")
fd(lambda t: m._gn("layers.3.gn1", t, 3), x8b, "gn+relu C=32")
fd(lambda t: m._gn("layers.3.gn2", t, 0), x8, "gn C=64")
x4 = torch.randn(4, 4, 4, 128, device="cuda")
fd(lambda t: m._gn("layers.4.gn2", t, 0), x4, "gn C=128 P=16")

# exact check of one block against torch autograd on the same device (debug only)
import torch.nn.functional as F


def weff(prefix):
    W = m.P[prefix + ".weight_orig"]
    Wm = W.view(W.shape[0], -1)
    sigma = torch.dot(m.P[prefix + ".weight_u"], Wm @ m.P[prefix + ".weight_v"])
    return W / sigma


def tblock(b, x_nchw):
    p = f"layers.{b['i']}"
    g = lambda n: (m.P[p + n + ".weight"], m.P[p + n + ".bias"])
    o = F.relu(F.group_norm(F.conv2d(x_nchw, weff(p + ".conv1")), min(32, b["mid"]), *g(".gn1")))
    o = F.group_norm(F.conv2d(o, weff(p + ".conv2"), None, b["stride"], 1), 32, *g(".gn2"))
    if b["se"]:
        y = o.mean(dim=(2, 3))
        y = F.relu(F.linear(y, m.P[p + ".se.fc1.weight"], m.P[p + ".se.fc1.bias"]))
        y = torch.sigmoid(F.linear(y, m.P[p + ".se.fc2.weight"], m.P[p + ".se.fc2.bias"]))
        o = o * y[:, :, None, None]
    sk = F.group_norm(F.conv2d(x_nchw, weff(p + ".skip.0"), None, b["stride"]), 32, *g(".skip.1")) if b["skip"] else x_nchw
    return F.relu(o + sk)


for b in m.blocks:
    side = {0: 16, 1: 16, 2: 8, 3: 8, 4: 4}[b["i"]]
    x = torch.randn(4, side, side, b["cin"], device="cuda")
    out, bwd = m._block(b, x)
    R = torch.randn_like(out)
    m.grads = {}
    dx = bwd(R)
    xt = x.permute(0, 3, 1, 2).contiguous().requires_grad_()
    ot = tblock(b, xt)
    ot.backward(R.permute(0, 3, 1, 2).contiguous())
    e_out = float((out.permute(0, 3, 1, 2) - ot).abs().max() / ot.abs().max())
    e_dx = float((dx.permute(0, 3, 1, 2) - xt.grad).abs().max() / xt.grad.abs().max())
    print(f"block {b['i']}: forward err {e_out:.2e}  dx err {e_dx:.2e}")

print("training mode (power iteration inside _block; torch side uses the updated u, v):")
m.train()
for b in m.blocks:
    side = {0: 16, 1: 16, 2: 8, 3: 8, 4: 4}[b["i"]]
    x = torch.randn(4, side, side, b["cin"], device="cuda")
    out, bwd = m._block(b, x)
    R = torch.randn_like(out)
    m.grads = {}
    dx = bwd(R)
    xt = x.permute(0, 3, 1, 2).contiguous().requires_grad_()
    ot = tblock(b, xt)
    ot.backward(R.permute(0, 3, 1, 2).contiguous())
    e_out = float((out.permute(0, 3, 1, 2) - ot).abs().max() / ot.abs().max())
    e_dx = float((dx.permute(0, 3, 1, 2) - xt.grad).abs().max() / xt.grad.abs().max())
    print(f"block {b['i']}: forward err {e_out:.2e}  dx err {e_dx:.2e}")

print("chained trunk (eval mode): gradient at every block boundary vs torch autograd")
m.eval()
x = torch.rand(4, 32 * 32, device="cuda")
x4 = x.view(4, 32, 32, 1).contiguous()
c0, bw0 = m._conv("initial_conv.0", x4, 7, 1, 3, need_dx=False)
a0, bwg = m._gn("initial_conv.1", c0, 3)
from simulgen_vae_amd import ops
h, _pool_idx = ops.maxpool_fwd(a0)
mine = [h]
bws = []
for b in m.blocks:
    h, bw = m._block(b, h)
    mine.append(h)
    bws.append(bw)
R = torch.randn_like(h)
xt = x.view(4, 1, 32, 32)
t = F.max_pool2d(F.relu(F.group_norm(F.conv2d(xt, weff("initial_conv.0"), None, 1, 3), 16, m.P["initial_conv.1.weight"], m.P["initial_conv.1.bias"])), 3, 2, 1)
ts = [t.detach().requires_grad_()]
cur = ts[0]
outs = []
for b in m.blocks:
    cur = tblock(b, cur)
    cur.retain_grad()
    outs.append(cur)
print("stem forward err", float((mine[0].permute(0, 3, 1, 2) - t).abs().max() / t.abs().max()))
cur.backward(R.permute(0, 3, 1, 2).contiguous())
d = R
m.grads = {}
for i in range(len(m.blocks) - 1, -1, -1):
    d = bws[i](d)
    ref = (outs[i - 1].grad if i > 0 else ts[0].grad)
    print(f"  grad at input of block {i}: err {float((d.permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max()):.2e}   forward err of block {i} output {float((mine[i + 1].permute(0, 3, 1, 2) - outs[i]).abs().max() / outs[i].abs().max()):.2e}")
