set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest6.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest6.log
tail -5 gpurun_out/r02/pytest6.log
CORRLA_SVD=mc timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or widths or rank_deficient or scale_invariance or sizes_cross or svd_paths or sign_convention or fuzz or sweep" > gpurun_out/r02/pytest6_mc.log 2>&1; echo "pytest mc rc=$?" | tee -a gpurun_out/r02/pytest6_mc.log
tail -5 gpurun_out/r02/pytest6_mc.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_c2_f.json 2> gpurun_out/r02/bench_c2_f.err; echo "bench rc=$?"
tail -4 gpurun_out/r02/bench_c2_f.err
CORRLA_JMC_MIN_L=100 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_c2_f_mc.json 2> gpurun_out/r02/bench_c2_f_mc.err; echo "bench rc=$?"
tail -3 gpurun_out/r02/bench_c2_f_mc.err
timeout -k 10 400 python bench.py --config C4 --steps 5 --warmup 2 > gpurun_out/r02/bench_c4_f.json 2> gpurun_out/r02/bench_c4_f.err; echo "bench c4 rc=$?"
tail -4 gpurun_out/r02/bench_c4_f.err
timeout -k 10 300 python tools/bench_configs.py C3q2 C3 C4shard > gpurun_out/r02/configs_f.jsonl 2> gpurun_out/r02/configs_f.err; cat gpurun_out/r02/configs_f.jsonl
CORRLA_MW=1 timeout -k 10 300 python tools/bench_configs.py C4shard 2>/dev/null
timeout -k 10 300 python tools/debug_orth.py 2>&1 | grep -v amdgpu.ids | grep "{}"
