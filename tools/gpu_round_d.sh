set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 300 python tools/debug_orth.py 2>&1 | grep -v amdgpu.ids
CORRLA_DEBUG=1 MODES=mc LS=138,266 timeout -k 10 300 python tools/bench_core_svd.py f32 f64 2>&1 | grep -E "flat|per step|multi-workgroup," | sort | uniq -c | sort -rn | head -20
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest4.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest4.log
tail -5 gpurun_out/r02/pytest4.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_c2_d.json 2> gpurun_out/r02/bench_c2_d.err; echo "bench rc=$?"
tail -4 gpurun_out/r02/bench_c2_d.err
timeout -k 10 400 python bench.py --config C4 --steps 5 --warmup 2 > gpurun_out/r02/bench_c4_d.json 2> gpurun_out/r02/bench_c4_d.err; echo "bench c4 rc=$?"
tail -4 gpurun_out/r02/bench_c4_d.err
timeout -k 10 300 python tools/bench_configs.py C3q2 C3 C4shard > gpurun_out/r02/configs_d.jsonl 2> gpurun_out/r02/configs_d.err; cat gpurun_out/r02/configs_d.jsonl
timeout -k 10 600 bash tools/collect_profiles.sh r02d tl_c4 > gpurun_out/r02/collect_d.log 2>&1
cat gpurun_out/prof_r02d/step_timeline_c4shard.txt | tail -62
