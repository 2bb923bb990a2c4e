set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest14.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest14.log
tail -8 gpurun_out/r02/pytest14.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/bench_gemm_nn.py 1250000 512 80 2>/dev/null && \
timeout -k 10 120 python tools/bench_gemm_nn.py 16384 16384 138 2>/dev/null && \
timeout -k 10 300 python tools/bench_configs.py C4shard C2 C3q2 C4full 2>/dev/null | cut -c1-800 | tee gpurun_out/r02/configs_o.jsonl
