"""Debug aid: the whole LatentConditionerImg mirror (training mode, injected dropout masks) against torch autograd
of a restatement on the same device.   python tests/micro/lc_full_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, torch.nn.functional as F
import simulgen_vae_amd
from simulgen_vae_amd.modules.latent_conditioner_model_cnn import LatentConditionerImg, _num_groups
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "lc_small.npz"))
m = LatentConditionerImg([int(v) for v in g["filters"]], 32, (1, 32, 32), 8, 3, (32, 32), compute_dtype="f32")
m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("s0.")})
m.train()
rec = {}
orig_block = m._block


def wrapped(b, x4):
    out, bwd = orig_block(b, x4)

    def bw(dout):
        rec[("dout", b["i"])] = dout.clone()
        dx = bwd(dout)
        rec[("dx", b["i"])] = dx.clone()
        return dx
    rec[("out", b["i"])] = out
    return out, bw


m._block = wrapped
masks = [torch.from_numpy(g[f"mask{i}"]).cuda() for i in range(7)]
loss, A, Bl = m.loss_backward(torch.from_numpy(g["x"]), g["y1"], g["y2"], dropout_masks=list(masks))
print("loss", loss, g["loss"][0])
L = {n: t.detach().clone().requires_grad_() for n, t in m.named_parameters()}
P = m.P


def weff(prefix):
    W = L[prefix + ".weight_orig"]
    return W / torch.dot(P[prefix + ".weight_u"], W.view(W.shape[0], -1) @ P[prefix + ".weight_v"])


def gnp(prefix):
    return L[prefix + ".weight"], L[prefix + ".bias"]


x = torch.from_numpy(g["x"]).cuda().view(4, 1, 32, 32)
h = F.max_pool2d(F.relu(F.group_norm(F.conv2d(x, weff("initial_conv.0"), None, 1, 3), _num_groups(16), *gnp("initial_conv.1"))), 3, 2, 1)
for b in m.blocks:
    p = f"layers.{b['i']}"
    o = F.relu(F.group_norm(F.conv2d(h, weff(p + ".conv1")), _num_groups(b["mid"]), *gnp(p + ".gn1")))
    o = F.group_norm(F.conv2d(o, weff(p + ".conv2"), None, b["stride"], 1), _num_groups(b["cout"]), *gnp(p + ".gn2"))
    if b["se"]:
        y = o.mean(dim=(2, 3))
        y = F.relu(F.linear(y, L[p + ".se.fc1.weight"], L[p + ".se.fc1.bias"]))
        y = torch.sigmoid(F.linear(y, L[p + ".se.fc2.weight"], L[p + ".se.fc2.bias"]))
        o = o * y[:, :, None, None]
    sk = F.group_norm(F.conv2d(h, weff(p + ".skip.0"), None, b["stride"]), _num_groups(b["cout"]), *gnp(p + ".skip.1")) if b["skip"] else h
    h = F.relu(o + sk)
    h.retain_grad()
    rec[("th", b["i"])] = h
f = h.mean(dim=(2, 3))
r = 0.3
mk = list(masks)
drop = lambda t, p: t * mk.pop(0) / (1 - p)
f = drop(f, r * 0.3)
f = F.relu(F.layer_norm(F.linear(f, weff("feature_processor.1"), L["feature_processor.1.bias"]), (256,), *gnp("feature_processor.2")))
f = drop(f, r * 0.4)
f = F.relu(F.layer_norm(F.linear(f, weff("feature_processor.5"), L["feature_processor.5.bias"]), (256,), *gnp("feature_processor.6")))
features = drop(f, r * 0.4)


def head(name, skipn, outn):
    t = F.linear(features, weff(name + "_layer1.0"), L[name + "_layer1.0.bias"])
    t = drop(F.relu(F.batch_norm(t, None, None, *gnp(name + "_layer1.1"), training=True)), r * 0.3)
    t = F.linear(t, weff(name + "_layer2.0"), L[name + "_layer2.0.bias"])
    t = drop(F.relu(F.batch_norm(t, None, None, *gnp(name + "_layer2.1"), training=True)), 0.2)
    return F.linear(t + F.linear(features, L[skipn + ".weight"], L[skipn + ".bias"]), L[outn + ".weight"], L[outn + ".bias"])


p1 = head("latent_main", "main_skip_proj", "latent_main_output")
p2 = head("xs", "xs_skip_proj", "xs_output")
lt = 10 * F.mse_loss(p1, torch.from_numpy(g["y1"]).cuda()) + F.mse_loss(p2, torch.from_numpy(g["y2"]).cuda().view(4, -1))
lt.backward()
print("torch restatement loss", float(lt))
errs = sorted(((float((m.grads[n] - L[n].grad).abs().max() / (L[n].grad.abs().max() + 1e-12)), n) for n in L if L[n].grad.abs().max() > 1e-5), reverse=True)
print("vs torch restatement on device:", errs[:5])
errs2 = sorted(((float((L[n].grad.cpu() - torch.from_numpy(g['g.' + n])).abs().max() / (np.abs(g['g.' + n]).max() + 1e-12)), n) for n in L if np.abs(g['g.' + n]).max() > 1e-5), reverse=True)
print("torch restatement vs golden:", errs2[:5])

for i in range(4, -1, -1):
    th = rec[("th", i)]
    e_f = float((rec[("out", i)].permute(0, 3, 1, 2) - th).abs().max() / th.abs().max())
    e_d = float((rec[("dout", i)].permute(0, 3, 1, 2) - th.grad).abs().max() / th.grad.abs().max())
    print(f"block {i}: forward err {e_f:.2e}   d(out) err {e_d:.2e}")

print("block 3 alone on the REAL input / upstream gradient of the full run (eval mode: same u, v as used above):")
m._block = orig_block
m.eval()
b3 = m.blocks[3]
xin = rec[("out", 2)].clone()
dout = rec[("dout", 3)].clone()
out, bwd = m._block(b3, xin)
m.grads = {}
dx = bwd(dout)
print("  forward vs full run:", float((out - rec[("out", 3)]).abs().max()), " dx vs full-run dx:", float((dx - rec[("dx", 3)]).abs().max() / rec[("dx", 3)].abs().max()),
      " dx vs torch:", float((dx.permute(0, 3, 1, 2) - rec[("th", 2)].grad).abs().max() / rec[("th", 2)].grad.abs().max()))
xr = torch.randn_like(xin)
out, bwd = m._block(b3, xr)
dxr = bwd(dout)

print("inside block 3 (real data): each gradient vs torch autograd")
p = "layers.3"
b = b3
xt = xin.permute(0, 3, 1, 2).contiguous().requires_grad_()
Lw = {n: P[n] for n in P}
def weff2(prefix):
    W = P[prefix + ".weight_orig"]
    return W / torch.dot(P[prefix + ".weight_u"], W.view(W.shape[0], -1) @ P[prefix + ".weight_v"])
tc1 = F.conv2d(xt, weff2(p + ".conv1")); tc1.retain_grad()
ta1 = F.relu(F.group_norm(tc1, 32, P[p + ".gn1.weight"], P[p + ".gn1.bias"])); ta1.retain_grad()
tc2 = F.conv2d(ta1, weff2(p + ".conv2"), None, 2, 1); tc2.retain_grad()
to2 = F.group_norm(tc2, 32, P[p + ".gn2.weight"], P[p + ".gn2.bias"]); to2.retain_grad()
ty = to2.mean(dim=(2, 3))
ty = F.relu(F.linear(ty, P[p + ".se.fc1.weight"], P[p + ".se.fc1.bias"]))
ty = torch.sigmoid(F.linear(ty, P[p + ".se.fc2.weight"], P[p + ".se.fc2.bias"]))
to3 = to2 * ty[:, :, None, None]; to3.retain_grad()
tsc = F.conv2d(xt, weff2(p + ".skip.0"), None, 2); tsc.retain_grad()
tsk = F.group_norm(tsc, 32, P[p + ".skip.1.weight"], P[p + ".skip.1.bias"]); tsk.retain_grad()
tout = F.relu(to3 + tsk)
tout.backward(dout.permute(0, 3, 1, 2).contiguous())
c1, bw_c1 = m._conv(p + ".conv1", xin, 1, 1, 0)
a1, bw_g1 = m._gn(p + ".gn1", c1, 3)
c2, bw_c2 = m._conv(p + ".conv2", a1, 3, 2, 1)
o2, bw_g2 = m._gn(p + ".gn2", c2, 0)
cmp = lambda mine, ref, nm: print(f"  {nm}: {float((mine.permute(0, 3, 1, 2) - ref).abs().max() / (ref.abs().max() + 1e-30)):.2e}")
cmp(c1, tc1, "fwd conv1"); cmp(a1, ta1, "fwd gn1+relu"); cmp(c2, tc2, "fwd conv2"); cmp(o2, to2, "fwd gn2")
from simulgen_vae_amd import ops as O
d_o2 = to2.grad.permute(0, 2, 3, 1).contiguous()
d_c2 = bw_g2(d_o2); cmp(d_c2, tc2.grad, "d conv2-out (gn2 bwd)")
d_a1 = bw_c2(tc2.grad.permute(0, 2, 3, 1).contiguous()); cmp(d_a1, ta1.grad, "d a1 (conv2 3x3 s2 dX)")
d_c1 = bw_g1(ta1.grad.permute(0, 2, 3, 1).contiguous()); cmp(d_c1, tc1.grad, "d conv1-out (gn1+relu bwd)")
sc, bw_sc = m._conv(p + ".skip.0", xin, 1, 2, 0)
sk, bw_sg = m._gn(p + ".skip.1", sc, 0)
d_sc = bw_sg(tsk.grad.permute(0, 2, 3, 1).contiguous()); cmp(d_sc, tsc.grad, "d skipconv-out (skip gn bwd)")
Bn, Hh, Ww, Cc = o2.shape
o2f = o2.view(Bn, Hh * Ww, Cc)
pooled = O.avgpool_fwd(o2f)
hid, bw_f1 = m._linear(p + ".se.fc1", pooled, False, O.LIN_RELU)
s_, bw_f2 = m._linear(p + ".se.fc2", hid, False, O.LIN_SIGMOID)
o3 = O.chan_scale_fwd(o2f, s_).view(o2.shape)
cmp(o3, to3, "fwd SE output")
print("  fwd s:", float((s_ - ty).abs().max()))
d3 = to3.grad.permute(0, 2, 3, 1).contiguous()
dx_scale, ds = O.chan_scale_bwd(o2f, s_, d3.view(Bn, Hh * Ww, Cc))
ty.retain_grad() if False else None
dpool = bw_f1(bw_f2(ds))
d_o2m = O.avgpool_bwd(dpool, dx_scale).view(o2.shape)
cmp(d_o2m, to2.grad, "d o2 (SE backward)")
outm = O.add_relu(o3, sk)
dm = O.relu_bwd(outm, dout)
cmp(dm, to3.grad, "d after relu gate (= d o3)")
cmp(dm, tsk.grad, "d after relu gate (= d skip)")
dx_main = bw_c1(tc1.grad.permute(0, 2, 3, 1).contiguous())
dx_skip = bw_sc(tsc.grad.permute(0, 2, 3, 1).contiguous())
# torch: separate contributions
g_main = torch.autograd.grad(F.conv2d(xt, weff2(p + ".conv1")), xt, tc1.grad)[0]
g_skip = torch.autograd.grad(F.conv2d(xt, weff2(p + ".skip.0"), None, 2), xt, tsc.grad)[0]
cmp(dx_main, g_main, "dX conv1 (1x1 direct)")
cmp(dx_skip, g_skip, "dX skip conv (1x1 s2)")
cmp(O.add(dx_main, dx_skip), xt.grad, "sum")
print("  torch: g_main + g_skip vs xt.grad", float((g_main + g_skip - xt.grad).abs().max() / xt.grad.abs().max()))

print("re-running the whole block through m._block vs step-by-step, same data:")
out_b, bwd_b = m._block(b3, xin)
m.grads = {}
dx_b = bwd_b(dout)
cmp(dx_b, xt.grad, "m._block dx")
# manual composition with the same closures order as _block
c1, bw_c1 = m._conv(p + ".conv1", xin, 1, 1, 0)
a1, bw_g1 = m._gn(p + ".gn1", c1, 3)
c2, bw_c2 = m._conv(p + ".conv2", a1, 3, 2, 1)
o2, bw_g2 = m._gn(p + ".gn2", c2, 0)
o2f = o2.view(Bn, Hh * Ww, Cc)
pooled = O.avgpool_fwd(o2f)
hid, bw_f1 = m._linear(p + ".se.fc1", pooled, False, O.LIN_RELU)
s_, bw_f2 = m._linear(p + ".se.fc2", hid, False, O.LIN_SIGMOID)
o3 = O.chan_scale_fwd(o2f, s_).view(o2.shape)
sc, bw_sc = m._conv(p + ".skip.0", xin, 1, 2, 0)
sk, bw_sg = m._gn(p + ".skip.1", sc, 0)
outm = O.add_relu(o3, sk)
d = O.relu_bwd(outm, dout)
cmp(d, to3.grad, "d")
dx_scale, ds = O.chan_scale_bwd(o2f, s_, d.view(Bn, Hh * Ww, Cc))
dpool = bw_f1(bw_f2(ds))
d_o2 = O.avgpool_bwd(dpool, dx_scale).view(o2.shape)
cmp(d_o2, to2.grad, "d_o2")
t1 = bw_g2(d_o2); cmp(t1, tc2.grad, "after gn2 bwd")
t2 = bw_c2(t1); cmp(t2, ta1.grad, "after conv2 dX")
t3 = bw_g1(t2); cmp(t3, tc1.grad, "after gn1 bwd")
t4 = bw_c1(t3); cmp(t4, g_main, "after conv1 dX")
u1 = bw_sg(d); cmp(u1, tsc.grad, "after skip gn bwd")
u2 = bw_sc(u1); cmp(u2, g_skip, "after skip conv dX")
cmp(O.add(t4, u2), xt.grad, "sum")

print("torch-only consistency: full-model gradient at block 2 output vs block 3 re-evaluated alone")
t2in = rec[("th", 2)].detach().clone().requires_grad_()
hh = t2in
bb = m.blocks[3]
pp = "layers.3"
oo = F.relu(F.group_norm(F.conv2d(hh, weff(pp + ".conv1")), _num_groups(bb["mid"]), *gnp(pp + ".gn1")))
oo = F.group_norm(F.conv2d(oo, weff(pp + ".conv2"), None, bb["stride"], 1), _num_groups(bb["cout"]), *gnp(pp + ".gn2"))
yy = oo.mean(dim=(2, 3))
yy = F.relu(F.linear(yy, L[pp + ".se.fc1.weight"], L[pp + ".se.fc1.bias"]))
yy = torch.sigmoid(F.linear(yy, L[pp + ".se.fc2.weight"], L[pp + ".se.fc2.bias"]))
oo = oo * yy[:, :, None, None]
ss = F.group_norm(F.conv2d(hh, weff(pp + ".skip.0"), None, bb["stride"]), _num_groups(bb["cout"]), *gnp(pp + ".skip.1"))
h3 = F.relu(oo + ss)
print("  forward vs full th3:", float((h3 - rec[("th", 3)]).abs().max()))
gg = torch.autograd.grad(h3, t2in, rec[("th", 3)].grad)[0]
print("  alone-with-leaves grad vs full th2.grad:", float((gg - rec[("th", 2)].grad).abs().max() / rec[("th", 2)].grad.abs().max()))
print("  xt.grad (P-based restatement) vs full th2.grad:", float((xt.grad - rec[("th", 2)].grad).abs().max() / rec[("th", 2)].grad.abs().max()))
print("  u of layers.3.conv1 now vs L-time:", float((P["layers.3.conv1.weight_u"] - torch.from_numpy(g["s1.layers.3.conv1.weight_u"]).cuda()).abs().max()))
print("  weight_orig now vs leaf:", float((P["layers.3.conv1.weight_orig"] - L["layers.3.conv1.weight_orig"]).abs().max()))

print("degenerate GroupNorm groups in block 3 (real data):")
v1 = tc1.detach().var(dim=(2, 3), unbiased=False)          # [B, 32] per (sample, channel) = per group (Cg = 1)
print("  conv1-out per-group variance: min %.3e  median %.3e  #<1e-5: %d of %d" % (float(v1.min()), float(v1.median()), int((v1 < 1e-5).sum()), v1.numel()))
tc2g = tc2.detach().view(4, 32, 2, 4, 4).var(dim=(2, 3, 4), unbiased=False)
print("  conv2-out per-group variance: min %.3e  median %.3e  #<1e-5: %d of %d" % (float(tc2g.min()), float(tc2g.median()), int((tc2g < 1e-5).sum()), tc2g.numel()))
xin_t = xin.permute(0, 3, 1, 2)
print("  block input: abs max %.3e  fraction exactly zero %.3f" % (float(xin_t.abs().max()), float((xin_t == 0).float().mean())))
# sensitivity: same block, input perturbed by 1e-6 relative
xp = (xin * (1 + 1e-6 * torch.randn_like(xin))).contiguous()
_, bw_p = m._block(b3, xp)
m.grads = {}
dxp = bw_p(dout)
print("  my dx: change under a 1e-6 relative input perturbation: %.3e" % float((dxp - dx_b).abs().max() / dx_b.abs().max()))

print("inputs of the two torch evaluations:")
print("  x: max|xt - th2| = %.3e (max|th2| %.3e)" % (float((xt.detach() - rec[("th", 2)].detach()).abs().max()), float(rec[("th", 2)].abs().max())))
dref = rec[("th", 3)].grad
dmine = dout.permute(0, 3, 1, 2)
print("  dout: max|mine - th3.grad| = %.3e (max %.3e)" % (float((dmine - dref).abs().max()), float(dref.abs().max())))
gg2 = torch.autograd.grad(h3, t2in, dmine.contiguous(), retain_graph=True)[0] if False else None
t2b = rec[("th", 2)].detach().clone().requires_grad_()
def blk(hh):
    oo = F.relu(F.group_norm(F.conv2d(hh, weff2(pp + ".conv1")), 32, P[pp + ".gn1.weight"], P[pp + ".gn1.bias"]))
    oo = F.group_norm(F.conv2d(oo, weff2(pp + ".conv2"), None, 2, 1), 32, P[pp + ".gn2.weight"], P[pp + ".gn2.bias"])
    yy = oo.mean(dim=(2, 3))
    yy = F.relu(F.linear(yy, P[pp + ".se.fc1.weight"], P[pp + ".se.fc1.bias"]))
    yy = torch.sigmoid(F.linear(yy, P[pp + ".se.fc2.weight"], P[pp + ".se.fc2.bias"]))
    oo = oo * yy[:, :, None, None]
    ss = F.group_norm(F.conv2d(hh, weff2(pp + ".skip.0"), None, 2), 32, P[pp + ".skip.1.weight"], P[pp + ".skip.1.bias"])
    return F.relu(oo + ss)
for nm, xx, dd in (("th2 + th3.grad", t2b, dref), ("th2 + my dout", rec[("th", 2)].detach().clone().requires_grad_(), dmine.contiguous()),
                   ("my x + th3.grad", xin.permute(0, 3, 1, 2).contiguous().requires_grad_(), dref)):
    gx = torch.autograd.grad(blk(xx), xx, dd)[0]
    print(f"  P-based block with {nm}: vs full th2.grad {float((gx - rec[('th', 2)].grad).abs().max() / rec[('th', 2)].grad.abs().max()):.3e}")

print("additive-noise sensitivity of the torch block-3 gradient:")
base = rec[("th", 2)].detach()
for amp in (1e-7, 1e-6, 1e-5):
    xx = (base + amp * torch.randn_like(base)).requires_grad_()
    gx = torch.autograd.grad(blk(xx), xx, dref)[0]
    print(f"  noise {amp:.0e}: gradient change {float((gx - rec[('th', 2)].grad).abs().max() / rec[('th', 2)].grad.abs().max()):.3e}")
xx = base.clone()
xx[base == 0] = 1e-6
xx.requires_grad_()
gx = torch.autograd.grad(blk(xx), xx, dref)[0]
print(f"  zeros -> 1e-6: gradient change {float((gx - rec[('th', 2)].grad).abs().max() / rec[('th', 2)].grad.abs().max()):.3e}")
diff = (xt.detach() - base)
print("  my x vs th2: #entries differing %d of %d; where th2==0: mine nonzero %d; where mine==0: th2 nonzero %d" % (int((diff != 0).sum()), diff.numel(), int(((base == 0) & (xt.detach() != 0)).sum()), int(((xt.detach() == 0) & (base != 0)).sum())))
hh1 = F.conv2d(base, weff2(pp + ".conv1")); hh2 = F.conv2d(xt.detach(), weff2(pp + ".conv1"))
z1 = F.group_norm(hh1, 32, P[pp + ".gn1.weight"], P[pp + ".gn1.bias"]); z2 = F.group_norm(hh2, 32, P[pp + ".gn1.weight"], P[pp + ".gn1.bias"])
print("  gn1 pre-activation: max diff %.3e ; relu mask disagreements %d of %d" % (float((z1 - z2).abs().max()), int(((z1 > 0) != (z2 > 0)).sum()), z1.numel()))

print("path from th2 to my x:")
mx = xt.detach()
for a in (0.0, 1e-3, 1e-2, 0.1, 0.5, 1.0):
    xx = (base + a * (mx - base)).requires_grad_()
    out_ = blk(xx)
    gx = torch.autograd.grad(out_, xx, dref)[0]
    print(f"  alpha {a}: forward change {float((out_ - rec[('th', 3)]).abs().max()):.3e}  gradient change {float((gx - rec[('th', 2)].grad).abs().max() / rec[('th', 2)].grad.abs().max()):.3e}")
d = (mx - base)
print("  diff stats: mean %.3e  abs mean %.3e ; per-channel mean abs max %.3e" % (float(d.mean()), float(d.abs().mean()), float(d.mean(dim=(0, 2, 3)).abs().max())))

print("stage-wise: torch block at x=mx (alpha 1) vs x=(mx+base)/2")
def stages(hh):
    c1 = F.conv2d(hh, weff2(pp + ".conv1")); c1.retain_grad()
    z1 = F.group_norm(c1, 32, P[pp + ".gn1.weight"], P[pp + ".gn1.bias"]); z1.retain_grad()
    a1 = F.relu(z1); a1.retain_grad()
    c2 = F.conv2d(a1, weff2(pp + ".conv2"), None, 2, 1); c2.retain_grad()
    o2 = F.group_norm(c2, 32, P[pp + ".gn2.weight"], P[pp + ".gn2.bias"]); o2.retain_grad()
    yy = o2.mean(dim=(2, 3))
    yy = F.relu(F.linear(yy, P[pp + ".se.fc1.weight"], P[pp + ".se.fc1.bias"]))
    yy = torch.sigmoid(F.linear(yy, P[pp + ".se.fc2.weight"], P[pp + ".se.fc2.bias"]))
    o3 = o2 * yy[:, :, None, None]; o3.retain_grad()
    sc = F.conv2d(hh, weff2(pp + ".skip.0"), None, 2); sc.retain_grad()
    sk = F.group_norm(sc, 32, P[pp + ".skip.1.weight"], P[pp + ".skip.1.bias"]); sk.retain_grad()
    pre = o3 + sk; pre.retain_grad()
    out = F.relu(pre)
    return dict(c1=c1, z1=z1, a1=a1, c2=c2, o2=o2, o3=o3, sc=sc, sk=sk, pre=pre, out=out)
xa = mx.clone().requires_grad_(); xb = (0.5 * (mx + base)).requires_grad_()
sa, sb = stages(xa), stages(xb)
sa["out"].backward(dref); sb["out"].backward(dref)
for k in ("pre", "sk", "sc", "o3", "o2", "c2", "a1", "z1", "c1"):
    fa, fb = sa[k], sb[k]
    print(f"  {k}: value diff {float((fa - fb).abs().max()):.2e}  grad diff {float((fa.grad - fb.grad).abs().max() / fb.grad.abs().max()):.2e}  exact zeros in value: {int((fa == 0).sum())}/{int((fb == 0).sum())}")
print("  x grad diff", float((xa.grad - xb.grad).abs().max() / xb.grad.abs().max()))
