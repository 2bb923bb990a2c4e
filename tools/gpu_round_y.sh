set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 400 python tools/bench_decay.py f64 65536 4096 256 10 2>/dev/null | cut -c1-400 | tee gpurun_out/r02/decay_c3q10_robust.jsonl
CORRLA_DEVICE_ROBUST_QR=0 timeout -k 10 600 python tools/bench_decay.py f64 65536 4096 256 10 2>/dev/null | cut -c1-400 | tee gpurun_out/r02/decay_c3q10_old.jsonl
timeout -k 10 400 python tools/bench_decay.py f32 16384 16384 128 6 2>/dev/null | cut -c1-400 | tee gpurun_out/r02/decay_c2q6_robust.jsonl
