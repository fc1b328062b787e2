#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): tools/profile_gpu.sh for every bench workload, then the size
# sweep under rocprofv3 --kernel-trace --stats (one average duration per kernel instantiation).
#   tools/profile_all.sh <tag>
set -o pipefail
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$PWD}
for w in fft4096 fft4096_f64 real4096 fft16k spectrum16k spectrum256 peaks16k; do
  extra=""
  case $w in spectrum16k|peaks16k) extra="--batch 65536";; spectrum256) extra="--batch 1048576";; esac
  bash "$REPO/tools/profile_gpu.sh" "${TAG}_$w" --workload $w $extra > "$REPO/gpurun_out/prof_${TAG}_$w.log" 2>&1 || { echo "profile $w failed"; tail -5 "$REPO/gpurun_out/prof_${TAG}_$w.log"; exit 1; }
  echo "profiled $w"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/gpurun_out/prof_${TAG}_sweep" -- python3 "$REPO/tools/sweep.py" > "$REPO/gpurun_out/prof_${TAG}_sweep.log" 2>&1 || exit 1
echo "profiled sweep"
