import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import corrla_rs_amd as cr
ctx = cr.Context(0)
a = torch.empty((16384, 2048), dtype=torch.float32, device="cuda")
ctx.fill_normal(a, seed=1)
for _ in range(3):
    u, s, vt = ctx.rsvd(a, 128, 0, 10, seed=1, qr="householder")
torch.cuda.synchronize()
