#!/bin/bash
# Collects the rocprofv3 evidence of a round into gpurun_out/prof_$TAG (run on the MI355X box from the repo root):
#   kernel-trace + stats of the default bench, one-step timelines (C2, C4 shard, C3), PMC passes on the tall GEMMs
#   (f32 C2 sketch, f64 C3 sketch; counters in separate passes, never combined with other trace domains).
# Usage: bash tools/collect_profiles.sh r03 [what...]   what: bench tl_c2 tl_c4 tl_c3 pmc_f32 pmc_f64 (default: these) pmc_c4 pmc_x6 pmc_x3 pmc_c4_x6 pmc_as
set -u
TAG=${1:-r02}; shift || true
WHAT=${*:-bench tl_c2 tl_c4 tl_c3 pmc_f32 pmc_f64}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has bench; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $REPO/bench.py --steps 20 --warmup 3 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
  cp $(find $OUT/bench -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv 2>/dev/null
fi
if has tl_c2; then
  rocprofv3 --kernel-trace -d $OUT/tl_c2 -o tl -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $OUT/tl_c2.err
  python3 $REPO/tools/step_timeline.py $(find $OUT/tl_c2 -name '*.db' | head -1) > $OUT/step_timeline_c2.txt 2>&1
fi
if has tl_c4; then
  rocprofv3 --kernel-trace -d $OUT/tl_c4 -o tl -- python3 $REPO/tools/bench_configs.py C4shard > $OUT/tl_c4.json 2> $OUT/tl_c4.err
  python3 $REPO/tools/step_timeline.py $(find $OUT/tl_c4 -name '*.db' | head -1) --call 4 > $OUT/step_timeline_c4shard.txt 2>&1
fi
if has tl_c3; then
  rocprofv3 --kernel-trace -d $OUT/tl_c3 -o tl -- python3 $REPO/tools/bench_configs.py C3 > $OUT/tl_c3.json 2> $OUT/tl_c3.err
  python3 $REPO/tools/step_timeline.py $(find $OUT/tl_c3 -name '*.db' | head -1) --call 4 > $OUT/step_timeline_c3.txt 2>&1
  python3 $REPO/tools/step_timeline.py $(find $OUT/tl_c3 -name '*.db' | head -1) --stats > $OUT/kernel_stats_c3.txt 2>&1
fi
pmc() {  # $1 = f32|f64|c4 [$2 = bf16x6|bf16x3 -> summary pmc_mixed_<mode>_...]
  local dt=$1; local mx=${2:-}; local tag=$dt; [[ -n "$mx" ]] && tag=mixed_${mx}_$dt
  local sum=$OUT/pmc_${tag}_gemm_summary.txt; : > $sum
  local arg="$dt $mx"
  for pass in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
              "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" \
              "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    local name=${pass%%:*}; local ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_${tag}_$name -o $name -- python3 $REPO/tools/profile_sketch.py 40 $arg > $OUT/pmc_${tag}_$name.log 2>&1
    echo "## pass pmc_$name" >> $sum
    python3 $REPO/tools/summarize_pmc.py $(find $OUT/pmc_${tag}_$name -name '*counter_collection.csv' | head -1) gemm >> $sum 2>&1
    python3 $REPO/tools/summarize_pmc.py $(find $OUT/pmc_${tag}_$name -name '*counter_collection.csv' | head -1) slab_reduce >> $sum 2>&1
    python3 $REPO/tools/summarize_pmc.py $(find $OUT/pmc_${tag}_$name -name '*counter_collection.csv' | head -1) split_planes >> $sum 2>&1
  done
}
pmc_as() {  # gradient stage at 1e6 x 64: the k-NN scan and the fit kernel
  local sum=$OUT/pmc_active_ss_summary.txt; : > $sum
  for pass in "sq:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
              "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" \
              "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
    local name=${pass%%:*}; local ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_as_$name -o $name -- python3 $REPO/tools/bench_active_ss.py 1000000 > $OUT/pmc_as_$name.log 2>&1
    echo "## pass pmc_$name" >> $sum
    python3 $REPO/tools/summarize_pmc.py $(find $OUT/pmc_as_$name -name '*counter_collection.csv' | head -1) knn2_kernel >> $sum 2>&1
    python3 $REPO/tools/summarize_pmc.py $(find $OUT/pmc_as_$name -name '*counter_collection.csv' | head -1) grad_fit >> $sum 2>&1
  done
}
has pmc_as && pmc_as
has pmc_f32 && pmc f32
has pmc_f64 && pmc f64
has pmc_c4 && pmc c4
has pmc_x6 && pmc f32 bf16x6
has pmc_x3 && pmc f32 bf16x3
has pmc_c4_x6 && pmc c4 bf16x6
ls -la $OUT | head -40
