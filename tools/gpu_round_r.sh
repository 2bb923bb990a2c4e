set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
O=gpurun_out/r02/jmc_maxb.jsonl; : > $O
for b in 8 12 16 20 24; do
  echo "{\"max_b\": $b}" >> $O
  CORRLA_JMC_MAX_B=$b LS=138 MODES=mc timeout -k 10 120 python tools/bench_core_svd.py f32 2>/dev/null >> $O || exit 1
  CORRLA_JMC_MAX_B=$b LS=266 MODES=mc timeout -k 10 120 python tools/bench_core_svd.py f64 2>/dev/null >> $O || exit 1
done
cat $O
