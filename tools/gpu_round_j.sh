set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_sweep" > gpurun_out/r02/pytest10_fused.log 2>&1; echo "pytest fused rc=$?" | tee -a gpurun_out/r02/pytest10_fused.log
tail -12 gpurun_out/r02/pytest10_fused.log
timeout -k 10 300 python tools/bench_configs.py C4shard 2>/dev/null | tee gpurun_out/r02/f4_unfused.json
FUSED=1 timeout -k 10 300 python tools/bench_configs.py C4shard 2>/dev/null | tee gpurun_out/r02/f4_fused.json
FUSED=1 timeout -k 10 300 python tools/bench_configs.py C4full 2>/dev/null | tee -a gpurun_out/r02/f4_fused.json
timeout -k 10 300 python tools/bench_configs.py C4full 2>/dev/null | tee -a gpurun_out/r02/f4_unfused.json
timeout -k 10 300 python tools/bench_configs.py C3q2 2>/dev/null
