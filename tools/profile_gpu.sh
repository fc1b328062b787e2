#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC
# passes for bench.py, written under gpurun_out/prof_<tag>/ ; summaries are then
# reduced by tools/pmc_summary.py and copied into profiles/ by hand.
#   tools/profile_gpu.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also $*"
# 1. per-kernel time (no counters in this pass)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1 || exit 1
# 2./3. HBM bytes: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slot budget)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1 || exit 1
# 4. LDS conflicts + wave cycles
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1 || true
python3 "$REPO/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json" && cat "$OUT/summary.json"
