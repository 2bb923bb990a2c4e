set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
cat > /tmp/chol_t.py <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch, corrla_rs_amd as cr
ctx = cr.Context(0)
a = torch.empty((16384, 4096), dtype=torch.float64, device="cuda")
ctx.fill_normal(a, seed=3)
for k in (118, 126, 128, 130, 134):
    print("l =", k + 10, file=sys.stderr, flush=True)
    for i in range(2):
        u, s, vt = ctx.rsvd(a, k, 2, 10, seed=1)
PY
CORRLA_DEVICE_ROBUST_QR=0 CORRLA_DEBUG=2 timeout -k 10 200 python /tmp/chol_t.py 2>&1 | cut -c1-200 > gpurun_out/r02/chol_times.txt || true
