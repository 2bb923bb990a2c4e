set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest12.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest12.log
tail -8 gpurun_out/r02/pytest12.log
[ $rc -eq 0 ] && timeout -k 10 300 python tools/bench_configs.py C4shard C2 C3q2 2>/dev/null | tee gpurun_out/r02/persist_on.jsonl && \
CORRLA_GEMM_PERSIST_TILES=0 timeout -k 10 300 python tools/bench_configs.py C4shard C2 C3q2 2>/dev/null | tee gpurun_out/r02/persist_off.jsonl
