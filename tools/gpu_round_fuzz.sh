set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest25.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest25.log
grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest25.log | head -30
timeout -k 10 800 python tools/fuzz_parity.py 1200 11 > gpurun_out/r02/fuzz_a.txt 2>&1; echo "rc=$?" >> gpurun_out/r02/fuzz_a.txt
tail -6 gpurun_out/r02/fuzz_a.txt
timeout -k 10 800 python tools/fuzz_parity.py 1200 12 > gpurun_out/r02/fuzz_b.txt 2>&1; echo "rc=$?" >> gpurun_out/r02/fuzz_b.txt
tail -6 gpurun_out/r02/fuzz_b.txt
timeout -k 10 300 python tools/bench_configs.py C5 C3 C2 2>/dev/null | cut -c1-420
