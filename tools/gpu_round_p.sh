set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
O=gpurun_out/r02/nt_ab.jsonl; : > $O
for rep in 1 2; do
  for lib in "" build_ab/libcorrla_rsvd_dmant.so build_ab/libcorrla_rsvd_storent.so build_ab/libcorrla_rsvd_both.so; do
    echo "{\"lib\": \"$lib\"}" >> $O
    CORRLA_RSVD_LIB=$lib timeout -k 10 120 python tools/bench_gemm_nn.py 1250000 512 80 2>/dev/null >> $O || exit 1
    CORRLA_RSVD_LIB=$lib timeout -k 10 120 python tools/bench_configs.py C2 C4shard 2>/dev/null | cut -c1-330 >> $O || exit 1
  done
done
cat $O
