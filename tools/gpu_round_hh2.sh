#!/bin/bash
# round 2, blocked (compact-WY, MFMA) Householder panels: tests, then timings against the unblocked panels
mkdir -p gpurun_out/r02hh2
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "householder" > gpurun_out/r02hh2/tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r02hh2/tests.log
tail -25 gpurun_out/r02hh2/tests.log
if [ $rc -ne 0 ]; then exit 0; fi
for wy in 1 0; do
  echo "== CORRLA_HH_WY=$wy" >> gpurun_out/r02hh2/c2.log
  CORRLA_HH_WY=$wy CORRLA_QR=householder timeout -k 10 300 python tools/bench_configs.py C2 C4shard C3q2 C1 >> gpurun_out/r02hh2/c2.log 2>&1
done
python - <<'PY'
import json
for line in open('gpurun_out/r02hh2/c2.log'):
    line=line.strip()
    if line.startswith('=='): print(line)
    elif line.startswith('{'):
        d=json.loads(line); print(d['config'], d['ms'], d['phases'].get('qr_ms'), d['orthU'])
PY
