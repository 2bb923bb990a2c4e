set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest24.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest24.log
grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest24.log | head -30
[ $rc -eq 0 ] || exit 1
LS=64,138,266 MODES=default timeout -k 10 200 python tools/bench_core_svd.py f64 2>/dev/null | cut -c1-200 && \
timeout -k 10 300 python tools/bench_configs.py C1 C3q2 C5 2>/dev/null | cut -c1-420
