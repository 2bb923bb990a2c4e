#!/bin/bash
mkdir -p gpurun_out/r02fz
for c in "62 23" "61 757" "62 373"; do
  for wy in 1 0; do
    echo "== $c WY=$wy"
    CORRLA_HH_WY=$wy timeout -k 10 120 python tools/fuzz_replay.py $c 2>&1 | grep -v amdgpu.ids | tail -3
  done
done
