#!/bin/bash
# round 2, Householder any-width / sharded: tests, then the C2 step with the Householder thin-Q for several block widths
mkdir -p gpurun_out/r02hh
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "householder" > gpurun_out/r02hh/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r02hh/tests.log
tail -5 gpurun_out/r02hh/tests.log
for w in 138 69 46 35 28; do
  echo "== CORRLA_HH_BLOCK=$w" >> gpurun_out/r02hh/c2.log
  CORRLA_HH_BLOCK=$w CORRLA_QR=householder timeout -k 10 300 python tools/bench_configs.py C2 C4shard >> gpurun_out/r02hh/c2.log 2>&1
done
echo "== C3q2 householder" >> gpurun_out/r02hh/c2.log
CORRLA_QR=householder timeout -k 10 300 python tools/bench_configs.py C3q2 >> gpurun_out/r02hh/c2.log 2>&1
grep -E "^==|\"ms\"|ms_per|config" gpurun_out/r02hh/c2.log | cut -c1-400
