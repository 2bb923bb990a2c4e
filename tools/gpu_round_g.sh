set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_sweep" > gpurun_out/r02/pytest7_fused.log 2>&1; echo "pytest fused rc=$?" | tee -a gpurun_out/r02/pytest7_fused.log
tail -30 gpurun_out/r02/pytest7_fused.log
timeout -k 10 300 python tools/bench_configs.py C4shard 2>/dev/null
FUSED=1 timeout -k 10 300 python tools/bench_configs.py C4shard 2>/dev/null
timeout -k 10 400 python bench.py --config C4 --steps 5 --warmup 2 --fused > gpurun_out/r02/bench_c4_g_fused.json 2> gpurun_out/r02/bench_c4_g_fused.err; echo "bench c4 fused rc=$?"
tail -4 gpurun_out/r02/bench_c4_g_fused.err
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest7.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest7.log
tail -5 gpurun_out/r02/pytest7.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_c2_g.json 2> gpurun_out/r02/bench_c2_g.err; echo "bench rc=$?"
tail -4 gpurun_out/r02/bench_c2_g.err
