set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest3.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest3.log
tail -15 gpurun_out/r02/pytest3.log
CORRLA_SVD=mc timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or widths or rank_deficient or scale_invariance or sizes_cross or svd_paths or sign_convention or fuzz or sweep" > gpurun_out/r02/pytest3_mc.log 2>&1; echo "pytest mc rc=$?" | tee -a gpurun_out/r02/pytest3_mc.log
tail -5 gpurun_out/r02/pytest3_mc.log
CORRLA_DEBUG=1 MODES=mc LS=138,200,266 timeout -k 10 300 python tools/bench_core_svd.py 2>&1 | grep -E "flat|corrla" | tail -30
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_c2_c.json 2> gpurun_out/r02/bench_c2_c.err; echo "bench rc=$?"
tail -4 gpurun_out/r02/bench_c2_c.err
timeout -k 10 400 python bench.py --config C4 --steps 5 --warmup 2 > gpurun_out/r02/bench_c4_c.json 2> gpurun_out/r02/bench_c4_c.err; echo "bench c4 rc=$?"
tail -4 gpurun_out/r02/bench_c4_c.err
timeout -k 10 300 python tools/bench_configs.py C1 C3q2 C3 C4shard C5 > gpurun_out/r02/configs_c.jsonl 2> gpurun_out/r02/configs_c.err; cat gpurun_out/r02/configs_c.jsonl
timeout -k 10 900 bash tools/collect_profiles.sh r02c tl_c2 tl_c4 tl_c3 > gpurun_out/r02/collect_c.log 2>&1; tail -5 gpurun_out/r02/collect_c.log
cat gpurun_out/prof_r02c/step_timeline_c4shard.txt | tail -70
