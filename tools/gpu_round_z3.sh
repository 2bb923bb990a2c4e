set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest21.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest21.log
grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest21.log | head -30
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_decay.py f64 65536 4096 256 2 2>/dev/null | cut -c1-420 | tee gpurun_out/r02/decay_c3_robust.jsonl && \
CORRLA_DEVICE_ROBUST_QR=0 timeout -k 10 300 python tools/bench_decay.py f64 65536 4096 256 2 2>/dev/null | cut -c1-420 | tee gpurun_out/r02/decay_c3_old.jsonl && \
timeout -k 10 300 python tools/bench_configs.py C3q2 C3 C2x4 2>/dev/null | cut -c1-520
