mkdir -p gpurun_out/r02b
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --share-gpu > gpurun_out/r02b/torchrun_share2.json 2> gpurun_out/r02b/torchrun_share2.err; echo rc=$?
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --share-gpu --dist-backend nccl --no-cpu-baseline > gpurun_out/r02b/torchrun_nccl_share2.json 2> gpurun_out/r02b/torchrun_nccl_share2.err; echo rc_nccl=$?
timeout -k 10 200 python bench.py --gpus 2 --steps 5 --warmup 2 --share-gpu --dist-backend nccl --no-cpu-baseline > gpurun_out/r02b/self_nccl_share2.json 2> gpurun_out/r02b/self_nccl_share2.err; echo rc_self=$?
python - <<'PY'
import json
for f in ("torchrun_share2","torchrun_nccl_share2","self_nccl_share2"):
    try:
        d=json.loads(open("gpurun_out/r02b/"+f+".json").read().strip().splitlines()[-1]); print(f, d["n_gpus"], round(d["value"],1), d.get("per_rank_roofline_frac"), json.dumps(d.get("gather"))[:420])
    except Exception as e:
        print(f, "no line:", e); print(open("gpurun_out/r02b/"+f+".err").read()[-1500:])
PY
