#!/bin/bash
# One parametrised runner for every batch sent to the MI355X box (replaces the per-session gpu_round_*.sh scripts):
#   gpurun --timeout T -- 'bash tools/gpu_session.sh tools/plans/NAME.plan [OUTDIR]'
# A plan is a text file of steps, one per line:   label | timeout_seconds | command
# (blank lines and lines starting with # are skipped).  Every step runs from the repo root under `timeout -k 10`, its
# stdout + stderr go to gpurun_out/OUTDIR/label.log (OUTDIR defaults to the plan's name), and the last lines are echoed
# so that gpurun's tail shows progress.  A step that is killed at its limit (exit 124 / 137) ends the session: after a
# GPU step times out nothing else is started in the same call.  A failing step (any other non-zero exit) is reported
# and the session goes on.  Before the first step the library is checked (tests/test_abi.py: loads, exports every symbol, the
# device code object holds every kernel the host code launches): a broken build ends the session in seconds instead of
# aborting every step (SKIP_PREFLIGHT=1 in the environment skips it, e.g. to time a build that fails the ISA guard).  tools/plans/standard.plan = the round-end sequence (GPU tests, smoke, default bench);
# plans of one-off experiments are not tracked (tools/plans/.gitignore).
set -u
PLAN=${1:?usage: gpu_session.sh PLAN [OUTDIR]}
NAME=$(basename "$PLAN" .plan)
OUT=gpurun_out/${2:-$NAME}
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/session.log"
if [[ -z "${SKIP_PREFLIGHT:-}" ]] && ! timeout -k 10 300 python3 -m pytest tests/test_abi.py -x -q > "$OUT/preflight.log" 2>&1; then
  echo "!! preflight failed: the library in this snapshot is broken, nothing was run" | tee -a "$OUT/session.log"
  tail -n 25 "$OUT/preflight.log"
  exit 1
fi
while IFS='|' read -r label tmo cmd; do
  label=$(echo "$label" | xargs); tmo=$(echo "$tmo" | xargs)
  [[ -z "$label" || "$label" == \#* ]] && continue
  echo "== [$label] (limit ${tmo}s): $cmd" | tee -a "$OUT/session.log"
  t0=$(date +%s)
  OUT="$OUT" timeout -k 10 "$tmo" bash -o pipefail -c "$cmd" > "$OUT/$label.log" 2>&1
  rc=$?
  echo "   rc=$rc after $(( $(date +%s) - t0 ))s" | tee -a "$OUT/session.log"
  grep -v "amdgpu.ids" "$OUT/$label.log" | tail -n 6
  if [[ $rc -eq 124 || $rc -eq 137 ]]; then
    echo "!! [$label] hit its limit: no further GPU step in this call" | tee -a "$OUT/session.log"
    exit 0
  fi
done < "$PLAN"
echo "== session done" | tee -a "$OUT/session.log"
