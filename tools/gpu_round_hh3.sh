#!/bin/bash
# per-kernel times of the Householder thin-Q (16384 x 138 f32), blocked panels
mkdir -p gpurun_out/r02hh3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02hh3/prof -- python3 $GRAFT_REPO_ROOT/tools/profile_tsqr.py > $GRAFT_REPO_ROOT/gpurun_out/r02hh3/run.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r02hh3/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
