set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest8.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest8.log
tail -5 gpurun_out/r02/pytest8.log
timeout -k 10 300 python tools/bench_configs.py C4shard C3q2 C3 2>/dev/null
FUSED=1 timeout -k 10 300 python tools/bench_configs.py C4shard 2>/dev/null
CORRLA_EVEN_BLOCKS=1 timeout -k 10 300 python tools/bench_configs.py C3q2 2>/dev/null
timeout -k 10 900 bash tools/collect_profiles.sh r02h pmc_f64 pmc_f32 > gpurun_out/r02/collect_h.log 2>&1
cat gpurun_out/prof_r02h/pmc_f64_gemm_summary.txt
cat gpurun_out/prof_r02h/pmc_f32_gemm_summary.txt
