#!/bin/bash
# Runs ON THE GPU BOX: SQ-level counters for one bench workload (two PMC passes, 8 SQ slots each).
#   tools/pmc_sq.sh <tag> [bench args...]
TAG=${1:-sq}; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 1 --ramp-seconds 0.1 --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p1" -- $BENCH > "$OUT/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/p2" -- $BENCH > "$OUT/p2.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pdsp" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:24s} {sum(x)/len(x):16.0f}")
PY
