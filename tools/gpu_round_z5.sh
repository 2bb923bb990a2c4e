set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest23.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest23.log
grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest23.log | head -30
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_default_z5.json 2> gpurun_out/r02/bench_default_z5.err || exit 1
cut -c1-260 gpurun_out/r02/bench_default_z5.json
timeout -k 10 300 python bench.py --config C4 --no-cpu-baseline > gpurun_out/r02/bench_c4_z5.json 2> gpurun_out/r02/bench_c4_z5.err || exit 1
cut -c1-260 gpurun_out/r02/bench_c4_z5.json
rocprofv3 --kernel-trace -d gpurun_out/prof_z5 -o tl -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> gpurun_out/r02/tl_z5.err
python3 tools/step_timeline.py $(find gpurun_out/prof_z5 -name '*.db' | head -1) > gpurun_out/r02/step_timeline_c2_z5.txt 2>&1
rm -rf gpurun_out/prof_z5
tail -2 gpurun_out/r02/step_timeline_c2_z5.txt
