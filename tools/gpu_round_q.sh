set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest15.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest15.log
tail -8 gpurun_out/r02/pytest15.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_configs.py C5 C4shard 2>/dev/null | cut -c1-800 | tee gpurun_out/r02/configs_q.jsonl
