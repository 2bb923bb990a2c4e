set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "not baseline_configs" > gpurun_out/r02/pytest2.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest2.log
tail -15 gpurun_out/r02/pytest2.log
CORRLA_SVD=mc timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or widths or rank_deficient or scale_invariance or sizes_cross or svd_paths or sign_convention or fuzz or sweep" > gpurun_out/r02/pytest2_mc.log 2>&1; echo "pytest mc rc=$?" | tee -a gpurun_out/r02/pytest2_mc.log
tail -15 gpurun_out/r02/pytest2_mc.log
timeout -k 10 300 python tools/bench_core_svd.py > gpurun_out/r02/core_svd_1.jsonl 2> gpurun_out/r02/core_svd_1.err; echo "rc=$?"
cat gpurun_out/r02/core_svd_1.jsonl
for np in 2 3 4 6; do CORRLA_JMC_NP=$np LS=138 MODES=mc timeout -k 10 120 python tools/bench_core_svd.py f32 | grep flat; done
for np in 6 9 12; do CORRLA_JMC_NP=$np LS=266 MODES=mc timeout -k 10 120 python tools/bench_core_svd.py f64 | grep flat; done
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_c2_b.json 2> gpurun_out/r02/bench_c2_b.err; echo "bench rc=$?"
tail -4 gpurun_out/r02/bench_c2_b.err
CORRLA_SVD=lds timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02/bench_c2_b_ring.json 2> gpurun_out/r02/bench_c2_b_ring.err; echo "bench rc=$?"
tail -3 gpurun_out/r02/bench_c2_b_ring.err
