set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
O=gpurun_out/r02/nn_ablation2.jsonl; : > $O
for d in 0 8 9 1; do
  CORRLA_GEMM_DEBUG=$d timeout -k 10 120 python tools/bench_gemm_nn.py 1250000 512 80 2>/dev/null >> $O || exit 1
  CORRLA_GEMM_PERSIST_TILES=0 CORRLA_GEMM_DEBUG=$d timeout -k 10 120 python tools/bench_gemm_nn.py 1250000 512 80 2>/dev/null >> $O || exit 1
done
cat $O
