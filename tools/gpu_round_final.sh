set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_default_final.json 2> gpurun_out/r02/bench_default_final.err || exit 1
cut -c1-300 gpurun_out/r02/bench_default_final.json
timeout -k 10 900 bash tools/collect_profiles.sh r02f bench tl_c2 tl_c4 tl_c3 pmc_f32 pmc_f64 > gpurun_out/r02/collect_f.log 2>&1 || exit 1
timeout -k 10 600 python tools/bench_configs.py C1 C2 C3q2 C3 C4shard C4full C5 C2col C2x4 2>/dev/null | cut -c1-900 > gpurun_out/r02/configs_final.jsonl || exit 1
cut -c1-160 gpurun_out/r02/configs_final.jsonl
timeout -k 10 300 python bench.py --config C4 --no-cpu-baseline > gpurun_out/r02/bench_c4_final.json 2> gpurun_out/r02/bench_c4_final.err || exit 1
timeout -k 10 200 python tools/bench_decay.py f32 2>/dev/null | cut -c1-600 > gpurun_out/r02/decay_f32_final.jsonl || exit 1
timeout -k 10 200 python tools/bench_decay.py f64 16384 8192 128 2 2>/dev/null | cut -c1-600 > gpurun_out/r02/decay_f64_final.jsonl || exit 1
find gpurun_out/prof_r02f -name '*.db' -delete; find gpurun_out/prof_r02f -size +2M -delete
