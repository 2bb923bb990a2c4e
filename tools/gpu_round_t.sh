set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest16.log 2>&1
rc=$?; echo "pytest rc=$rc" | tee -a gpurun_out/r02/pytest16.log
tail -8 gpurun_out/r02/pytest16.log
[ $rc -eq 0 ] || exit 1
O=gpurun_out/r02/jmc_lanes.jsonl; : > $O
LS=96,128,138,144 MODES=mc CORRLA_DEBUG=1 timeout -k 10 120 python tools/bench_core_svd.py f32 >> $O 2> gpurun_out/r02/jmc_lanes8.err || exit 1
echo '{"lanes16": 1}' >> $O
CORRLA_JMC_LANES16=1 LS=96,128,138,144 MODES=mc timeout -k 10 120 python tools/bench_core_svd.py f32 2>/dev/null >> $O || exit 1
LS=138,266 MODES=mc timeout -k 10 120 python tools/bench_core_svd.py f64 2>/dev/null >> $O || exit 1
cat $O
grep -h "jacobi_svd (multi" -A2 gpurun_out/r02/jmc_lanes8.err | tail -6
timeout -k 10 300 python tools/bench_configs.py C2 C3q2 2>/dev/null | cut -c1-700
