set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest18.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest18.log
grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest18.log | head -30
timeout -k 10 200 python tools/bench_decay.py f32 2>/dev/null | cut -c1-600 | tee gpurun_out/r02/decay_f32_x.jsonl && \
timeout -k 10 200 python tools/bench_decay.py f64 16384 8192 128 2 2>/dev/null | cut -c1-600 | tee gpurun_out/r02/decay_f64_x.jsonl
