set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 bash tools/collect_profiles.sh r02s bench tl_c2 tl_c4 tl_c3 pmc_f32 pmc_f64 > gpurun_out/r02/collect_s.log 2>&1 || exit 1
timeout -k 10 600 python tools/bench_configs.py C1 C2 C3q2 C3 C4shard C4full C5 C2col C2x4 2>/dev/null | cut -c1-900 > gpurun_out/r02/configs_s.jsonl || exit 1
cat gpurun_out/r02/configs_s.jsonl | cut -c1-200
timeout -k 10 300 python bench.py --config C4 --no-cpu-baseline > gpurun_out/r02/bench_c4_s.json 2> gpurun_out/r02/bench_c4_s.err || exit 1
cut -c1-400 gpurun_out/r02/bench_c4_s.json
find gpurun_out/prof_r02s -name '*.db' -delete; find gpurun_out/prof_r02s -size +2M -delete
