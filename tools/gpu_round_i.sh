set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest9.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r02/pytest9.log
tail -5 gpurun_out/r02/pytest9.log
timeout -k 10 300 python tools/bench_configs.py C3q2 C3 2>/dev/null
CORRLA_EVEN_BLOCKS=1 timeout -k 10 300 python tools/bench_configs.py C3q2 2>/dev/null
timeout -k 10 300 python tools/profile_sketch.py 40 f64
CORRLA_EVEN_BLOCKS=1 timeout -k 10 300 python tools/profile_sketch.py 40 f64
