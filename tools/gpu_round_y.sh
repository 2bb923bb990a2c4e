set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(timeout -k 10 200 python tools/fuzz_replay.py 51 426 wide; CORRLA_DEVICE_ROBUST_QR=0 timeout -k 10 200 python tools/fuzz_replay.py 51 426 wide) > gpurun_out/r02/replay.txt 2>&1 || true
cat gpurun_out/r02/replay.txt
