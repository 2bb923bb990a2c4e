import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import corrla_rs_amd as cr
from oracle import rsvd_oracle as orc
ctx = cr.Context(0)
np.set_printoptions(precision=4, linewidth=200, suppress=False)
for dtype in (np.float32, np.float64):
    a = orc.KNOWN_ANSWER_A.astype(dtype)
    for mode in ("lds", "mc"):
        os.environ["CORRLA_SVD"] = mode
        u, s, vt = ctx.rsvd(a, 5, 12, 10, seed=3)
        print(dtype.__name__, mode, "S", s.ravel())
        print("U", u); print("Vt", vt)
os.environ["CORRLA_SVD"] = "mc"
a = torch.empty((4096, 4096), dtype=torch.float32, device="cuda"); ctx.fill_normal(a, seed=3)
for l in (64, 138):
    ctx.rsvd(a, l - 10, 2, 10, seed=1)
a = a.double()
for l in (138, 266):
    ctx.rsvd(a, l - 10, 2, 10, seed=1)
