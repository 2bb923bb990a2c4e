#!/bin/bash
# end of round 2 after the Householder work: the whole GPU suite, smoke, a default bench line, the Householder option's timings
mkdir -p gpurun_out/r02z
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r02z/tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r02z/tests.log
tail -4 gpurun_out/r02z/tests.log
if [ $rc -ne 0 ]; then exit 0; fi
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02z/smoke.log 2>&1; tail -1 gpurun_out/r02z/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/r02z/bench_default.json 2> gpurun_out/r02z/bench_default.err
cut -c1-240 gpurun_out/r02z/bench_default.json
for wy in 1 0; do
  echo "== CORRLA_HH_WY=$wy" >> gpurun_out/r02z/hh.log
  CORRLA_HH_WY=$wy CORRLA_QR=householder timeout -k 10 300 python tools/bench_configs.py C1 C2 C3q2 C4shard >> gpurun_out/r02z/hh.log 2>&1
done
grep -c '"config"' gpurun_out/r02z/hh.log
