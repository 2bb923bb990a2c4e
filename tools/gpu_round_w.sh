set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
cat > /tmp/one_decay.py <<'PY'
import os, sys, json, time
sys.path.insert(0, os.getcwd())
import torch, corrla_rs_amd as cr
ctx = cr.Context(0)
g = torch.empty((16384, 16384), dtype=torch.float32, device="cuda")
ctx.fill_normal(g, seed=3)
a = g * (0.7 ** torch.arange(16384, device="cuda", dtype=torch.float32))
for i in range(3):
    u, s, vt = ctx.rsvd(a, 128, 2, 10, seed=1)
    print("call", i, ctx.timings()["total_ms"], file=sys.stderr, flush=True)
PY
CORRLA_DEBUG=1 timeout -k 10 200 python /tmp/one_decay.py 2>&1 | grep -E "thin-Q|call|status record|jacobi" | cut -c1-420 | tee gpurun_out/r02/decay07_debug.txt
