set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -q -k "grad or active or knn or c5 or C5" > gpurun_out/r02/pytest27.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "^FAILED|passed|failed|^E  " gpurun_out/r02/pytest27.log | head -20
timeout -k 10 300 python tools/fuzz_grad.py 100 3 2>&1 | tail -3
