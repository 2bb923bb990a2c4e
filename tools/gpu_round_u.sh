set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 200 python bench.py > gpurun_out/r02/bench_default_u.json 2> gpurun_out/r02/bench_default_u.err || exit 1
cut -c1-300 gpurun_out/r02/bench_default_u.json
timeout -k 10 200 python tools/bench_decay.py f32 2>/dev/null | cut -c1-600 | tee gpurun_out/r02/decay_f32.jsonl && \
timeout -k 10 200 python tools/bench_decay.py f64 16384 8192 128 2 2>/dev/null | cut -c1-600 | tee gpurun_out/r02/decay_f64.jsonl
