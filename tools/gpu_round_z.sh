set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest19.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest19.log
grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest19.log | head -30
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/bench_decay.py f64 16384 8192 128 2 2>/dev/null | cut -c1-420 && \
timeout -k 10 300 python tools/bench_configs.py C1 C2 C3q2 C3 C5 C4shard 2>/dev/null | cut -c1-600
