"""Is the conditioner's training forward reproducible from a restored state? (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import simulgen_vae_amd
from simulgen_vae_amd.modules.latent_conditioner_model_cnn import LatentConditionerImg
side = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = LatentConditionerImg([32, 64, 128, 256, 512, 1024], 32, (1, side, side), 8, 3, (side, side), dropout_rate=0.0, use_attention=True, compute_dtype="bf16")
m.train()
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.rand((16, side * side), generator=g, device="cuda")
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
outs = []
for r in range(3):
    m.load_state_dict(sd0)
    p1, p2 = m.forward(x)
    torch.cuda.synchronize()
    outs.append((p1.float().clone(), p2.float().clone()))
    m._tape = None
for r in range(1, 3):
    print("run", r, "max|dp1|", float((outs[r][0] - outs[0][0]).abs().max()), "max|p1|", float(outs[0][0].abs().max()),
          "max|dp2|", float((outs[r][1] - outs[0][1]).abs().max()))
