#!/bin/bash
# Runs ON THE GPU BOX: SQ counters (two passes of 8) for the N=16384 spectrum kernels of round 1
# (tools/kbench_er01, built from the round-1 tree: spectrum_split16k_kernel) and of this tree (tools/kbench:
# packed<13>, dif16k with a table / fused / rect window), same synthetic frames, one box.
#   tools/pmc_sq_kbench.sh <outdir-under-gpurun_out>
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/${1:-sq_kbench}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
export KB_SPLIT=1
for tag in r01 r02; do
  bin=$REPO/tools/kbench; [ $tag = r01 ] && bin=$REPO/tools/kbench_er01
  rocprofv3 --pmc $P1 --output-format csv -d "$OUT/${tag}_p1" -- $bin 16384 2 spec > "$OUT/${tag}_p1.log" 2>&1 || exit 1
  rocprofv3 --pmc $P2 --output-format csv -d "$OUT/${tag}_p2" -- $bin 16384 2 spec > "$OUT/${tag}_p2.log" 2>&1 || exit 1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for tag in ("r01", "r02"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{sys.argv[1]}/{tag}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "pdsp" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(f"[{tag}] {k}")
        wc = sum(v["SQ_WAVE_CYCLES"]) / max(1, len(v["SQ_WAVE_CYCLES"])) if "SQ_WAVE_CYCLES" in v else 0
        for c, x in sorted(v.items()):
            m = sum(x) / len(x)
            pct = f"  {100 * m / wc:5.1f} % of wave cycles" if wc and c.startswith(("SQ_WAIT", "SQ_ACTIVE")) else ""
            print(f"   {c:24s} {m:16.0f}{pct}")
PY
