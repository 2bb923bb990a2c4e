set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
CORRLA_DEBUG=1 MODES=mc LS=74,138,266 timeout -k 10 300 python tools/bench_core_svd.py f32 f64 > gpurun_out/r02/core_svd_e.log 2>&1; grep -v amdgpu gpurun_out/r02/core_svd_e.log | grep -A4 -E "^# " | grep -v "^--" | awk '!seen[$0]++' | head -120
timeout -k 10 300 python tools/debug_orth.py 2>&1 | grep -v amdgpu.ids | head -4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest5.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest5.log
tail -5 gpurun_out/r02/pytest5.log
CORRLA_SVD=mc timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or widths or rank_deficient or scale_invariance or sizes_cross or svd_paths or sign_convention or fuzz or sweep" > gpurun_out/r02/pytest5_mc.log 2>&1; echo "pytest mc rc=$?" | tee -a gpurun_out/r02/pytest5_mc.log
tail -5 gpurun_out/r02/pytest5_mc.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_c2_e.json 2> gpurun_out/r02/bench_c2_e.err; echo "bench rc=$?"
tail -4 gpurun_out/r02/bench_c2_e.err
CORRLA_JMC_MIN_L=100 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_c2_e_mc.json 2> gpurun_out/r02/bench_c2_e_mc.err; echo "bench rc=$?"
tail -3 gpurun_out/r02/bench_c2_e_mc.err
timeout -k 10 400 python bench.py --config C4 --steps 5 --warmup 2 > gpurun_out/r02/bench_c4_e.json 2> gpurun_out/r02/bench_c4_e.err; echo "bench c4 rc=$?"
tail -4 gpurun_out/r02/bench_c4_e.err
timeout -k 10 300 python tools/bench_configs.py C3q2 C3 C4shard > gpurun_out/r02/configs_e.jsonl 2> gpurun_out/r02/configs_e.err; cat gpurun_out/r02/configs_e.jsonl
