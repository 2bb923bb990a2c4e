set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "decaying" > gpurun_out/r02/pytest22.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest22.log
grep -E "^FAILED|passed|failed|^E  " gpurun_out/r02/pytest22.log | head -30
