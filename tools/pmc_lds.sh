#!/bin/bash
# Runs ON THE GPU BOX: LDS bank-conflict rate of the two headline kernels at a batch SMALL enough that the SQ
# counters do not saturate (at bench size SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE read exactly
# 2^24 / 2^26).  One rocprofv3 --pmc pass per workload, counters only.   tools/pmc_lds.sh <outdir> [rows] [frames]
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/${1:-gpurun_out/pmc_lds}
ROWS=${2:-2048}; FRAMES=${3:-512}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
CNT="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
B="--steps 3 --warmup 1 --ramp-seconds 0 --no-cpu-baseline --no-also --no-measure-traffic"
rocprofv3 --pmc $CNT --output-format csv -d "$OUT/fft4096" -- python3 $REPO/bench.py --workload fft4096 --batch $ROWS $B > "$OUT/fft4096.log" 2>&1
rocprofv3 --pmc $CNT --output-format csv -d "$OUT/spectrum16k" -- python3 $REPO/bench.py --workload spectrum16k --batch $FRAMES --chunk $FRAMES $B > "$OUT/spectrum16k.log" 2>&1
rocprofv3 --pmc $CNT --output-format csv -d "$OUT/fft4096_f64" -- python3 $REPO/tools/ab_real_packed.py --pmc-c2c-f64 $ROWS > "$OUT/fft4096_f64.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pdsp" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    print(k)
    for c in sorted(m):
        print(f"   {c:24s} {m[c]:16.0f}   (max over launches {max(v[c]):.0f})")
    if m.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   => bank-conflict cycles / LDS-array cycles = {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.4f}"
              f"   ; LDS-array cycles per LDS instruction = {m['SQ_LDS_IDX_ACTIVE'] / max(m.get('SQ_INSTS_LDS', 0), 1):.2f}")
PY
