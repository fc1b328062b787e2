#!/bin/bash
# Runs ON THE 1-GPU BOX: what one card can rehearse of the N > 1 path (both ranks share cuda:0).
#   1. the driver's launch form (torch.distributed.run), exchange over gloo on host copies
#   2. the self-launching form, RCCL requested: two ranks on one card -> RCCL refuses ("Duplicate GPU"), every rank
#      skips the leg together, gather.error in the line, the group (if any) destroyed, status 0
#   3. the RCCL self-test in a world of one (the collective itself, on the real planes)
# tools/rehearse_2ranks.sh <outdir>
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/${1:-gpurun_out/rehearsal}
mkdir -p "$OUT"; cd "$REPO"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --share-gpu > "$OUT/torchrun_share2_gloo.json" 2> "$OUT/torchrun_share2_gloo.err"; echo "rc torchrun/gloo = $?"
timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --share-gpu --dist-backend nccl --no-cpu-baseline --gather-timeout 60 > "$OUT/self_share2_rccl_refused.json" 2> "$OUT/self_share2_rccl_refused.err"; echo "rc self/rccl-on-one-card = $?"
timeout -k 10 300 python bench.py --rccl-selftest --steps 5 --warmup 2 --no-measure-traffic --no-also --no-cpu-baseline > "$OUT/rccl_selftest.json" 2> "$OUT/rccl_selftest.err"; echo "rc rccl-selftest = $?"
python - "$OUT" <<'PY'
import json, sys
for f in ("torchrun_share2_gloo", "self_share2_rccl_refused", "rccl_selftest"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/{f}.json").read().strip().splitlines()[-1])
        print(f, "n_gpus", d["n_gpus"], "value", round(d["value"], 1), "gather", json.dumps(d.get("gather"))[:600])
    except Exception as e:
        print(f, "no line:", e)
    err = open(f"{sys.argv[1]}/{f}.err").read()
    print("   stderr mentions destroy_process_group:", "destroy_process_group() was not called" in err, "| bytes of stderr:", len(err))
PY
