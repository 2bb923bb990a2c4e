set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q -k "tall or c4 or C4 or C5 or c5" > gpurun_out/r02/pytest26.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest26.log | head
[ $rc -eq 0 ] || exit 1
rocprofv3 --kernel-trace -d gpurun_out/prof_y -o tl -- python3 tools/bench_configs.py C4shard > gpurun_out/r02/c4_y.txt 2>&1
python3 tools/step_timeline.py $(find gpurun_out/prof_y -name '*.db' | head -1) --call 4 > gpurun_out/r02/tl_c4_y.txt 2>&1
rm -rf gpurun_out/prof_y
grep -E "tall_|step span" gpurun_out/r02/tl_c4_y.txt | cut -c1-120
timeout -k 10 300 python tools/bench_configs.py C4shard C5 2>/dev/null | cut -c1-330
