"""Is the conditioner's training step reproducible from a restored state? (diagnostic: two forwards + backwards with the same
dropout masks; prints the first tensors that differ)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simulgen_vae_amd
from simulgen_vae_amd.modules.latent_conditioner_model_cnn import LatentConditionerImg
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = 16
m = LatentConditionerImg([32, 64, 128, 256, 512, 1024], 32, (1, side, side), 8, 3, (side, side), dropout_rate=0.0, use_attention=True, compute_dtype="bf16")
m.train()
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand((B, side * side), generator=g, device="cuda")
y1 = torch.randn((B, 32), generator=g, device="cuda") * 0.3
y2 = torch.randn((B, 3, 8), generator=g, device="cuda") * 0.3
width = m.P["latent_main_layer2.0.bias"].shape[0]
masks = [(torch.rand((B, width), generator=g, device="cuda") >= 0.2).float() for _ in range(2)]
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
outs = []
for r in range(3):
    m.load_state_dict(sd0)
    m.grads = {}
    p1, p2 = m.forward(x, [t.clone() for t in masks])
    loss, A, Bv = m.loss_backward(x, y1, y2, preds=(p1, p2))
    torch.cuda.synchronize()
    outs.append((p1.float().clone(), p2.float().clone(), loss, {k: v.float().clone() for k, v in m.grads.items()},
                 {k: v.clone() for k, v in m.state_dict().items()}))
for r in range(1, 3):
    print("run", r, "max|dp1|", float((outs[r][0] - outs[0][0]).abs().max()), "max|dp2|", float((outs[r][1] - outs[0][1]).abs().max()),
          "loss", outs[r][2], outs[0][2])
    bad = [k for k in outs[0][3] if not torch.equal(outs[0][3][k], outs[r][3][k])]
    print("  gradients that differ:", len(bad), bad[:6])
    bad = [k for k in outs[0][4] if not torch.equal(outs[0][4][k].float(), outs[r][4][k].float())]
    print("  state entries that differ after the step:", len(bad), bad[:6])
