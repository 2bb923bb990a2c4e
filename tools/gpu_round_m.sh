set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q -k "tall or c4 or C4 or matmul" > gpurun_out/r02/pytest13.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest13.log
tail -15 gpurun_out/r02/pytest13.log
[ $rc -eq 0 ] || exit 1
tl() {  # name, env...
  local name=$1; shift
  env "$@" python tools/bench_configs.py C4shard 2>/dev/null | cut -c1-700 > gpurun_out/r02/var_$name.json
  cat gpurun_out/r02/var_$name.json
}
tl all && tl norotate CORRLA_GEMM_NO_ROTATE=1 && tl nopersist CORRLA_GEMM_PERSIST_TILES=0 && tl notall CORRLA_TALL_MIN_ROWS=0 && tl none CORRLA_TALL_MIN_ROWS=0 CORRLA_GEMM_PERSIST_TILES=0 CORRLA_GEMM_NO_ROTATE=1 && \
rocprofv3 --kernel-trace -d gpurun_out/prof_m/tl_c4 -o tl -- python3 tools/bench_configs.py C4shard > gpurun_out/prof_m_tl_c4.json 2> gpurun_out/prof_m_tl_c4.err && \
python3 tools/step_timeline.py $(find gpurun_out/prof_m/tl_c4 -name '*.db' | head -1) --call 4 > gpurun_out/r02/step_timeline_c4shard_m.txt 2>&1
rm -rf gpurun_out/prof_m
