set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
(for c in "61 757" "61 954" "62 373"; do timeout -k 10 200 python tools/fuzz_replay.py $c; CORRLA_DEVICE_ROBUST_QR=0 timeout -k 10 200 python tools/fuzz_replay.py $c; done; timeout -k 10 200 python tools/fuzz_replay.py 71 41 wide; CORRLA_DEVICE_ROBUST_QR=0 timeout -k 10 200 python tools/fuzz_replay.py 71 41 wide) > gpurun_out/r02/replay.txt 2>&1 || true
cat gpurun_out/r02/replay.txt
