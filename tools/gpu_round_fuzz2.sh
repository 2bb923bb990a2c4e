#!/bin/bash
# fuzz after the blocked Householder panels: a third of the cases run the Householder mode (narrow and wide sketches)
mkdir -p gpurun_out/r02fz
for sd in 61 62; do
  timeout -k 10 500 python tools/fuzz_parity.py 800 $sd > gpurun_out/r02fz/fuzz_$sd.txt 2>&1; echo "rc=$?" >> gpurun_out/r02fz/fuzz_$sd.txt
  tail -5 gpurun_out/r02fz/fuzz_$sd.txt
done
timeout -k 10 800 python tools/fuzz_parity.py 400 63 wide > gpurun_out/r02fz/fuzz_w63.txt 2>&1; echo "rc=$?" >> gpurun_out/r02fz/fuzz_w63.txt
tail -5 gpurun_out/r02fz/fuzz_w63.txt
