#!/bin/bash
# Runs ON THE GPU BOX: HBM write / read bytes per launch of spectrum_dif16k_kernel at several amplitude row pitches
# (separate rocprofv3 --pmc passes, counters only).  tools/pmc_amp_pitch.sh <outdir>
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/${1:-gpurun_out/pmc_pitch}
mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for P in 0 8224 -8193 -8224; do
  for C in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/p${P}_$C" -- python3 $REPO/tools/ab_amp_pitch.py --pmc $P > "$OUT/p${P}_$C.log" 2>&1
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    tag = f[len(sys.argv[1]):].strip("/").split("/")[0]
    for r in csv.DictReader(open(f)):
        if "spectrum_dif16k" in r["Kernel_Name"]:
            acc[(tag.split("_")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
alg_w, alg_r = 16384 * 8193 * 4, 16384 * 16384 * 4
for (tag, c), v in sorted(acc.items()):
    b = sum(v) / len(v) * 1024 * (2 if c == "FETCH_SIZE" else 1)  # KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)
    alg = alg_r if c == "FETCH_SIZE" else alg_w
    print(f"{tag:6s} {c:10s} {b/1e6:10.1f} MB per launch = {b/alg:.4f} x algorithmic ({len(v)} launches)")
PY
