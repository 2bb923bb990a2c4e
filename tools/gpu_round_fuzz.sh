set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 200 python tools/fuzz_replay.py 22 133 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest28.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^FAILED|passed|failed" gpurun_out/r02/pytest28.log | head
for sd in 22 41; do
  timeout -k 10 500 python tools/fuzz_parity.py 1000 $sd > gpurun_out/r02/fuzz_$sd.txt 2>&1; echo "rc=$?" >> gpurun_out/r02/fuzz_$sd.txt
  tail -4 gpurun_out/r02/fuzz_$sd.txt
done
for sd in 32 51; do
  timeout -k 10 800 python tools/fuzz_parity.py 500 $sd wide > gpurun_out/r02/fuzz_w$sd.txt 2>&1; echo "rc=$?" >> gpurun_out/r02/fuzz_w$sd.txt
  tail -4 gpurun_out/r02/fuzz_w$sd.txt
done
timeout -k 10 200 python tools/bench_decay.py f32 2>/dev/null | cut -c1-160
