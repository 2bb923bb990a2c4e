set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
nproc; free -g | head -2
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02/pytest1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest1.log
tail -30 gpurun_out/r02/pytest1.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_c2_a.json 2> gpurun_out/r02/bench_c2_a.err; echo "bench rc=$?"
tail -5 gpurun_out/r02/bench_c2_a.err
timeout -k 10 400 python bench.py --config C4 --steps 5 --warmup 2 > gpurun_out/r02/bench_c4_a.json 2> gpurun_out/r02/bench_c4_a.err; echo "bench c4 rc=$?"
tail -5 gpurun_out/r02/bench_c4_a.err
CORRLA_BENCH_FORCE_SHARDED=1 CORRLA_FORCE_ALLREDUCE=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_c2_sharded1.json 2> gpurun_out/r02/bench_c2_sharded1.err; echo "bench sharded rc=$?"
tail -3 gpurun_out/r02/bench_c2_sharded1.err
