#!/bin/bash
# round-2 GPU session 28: bench.py under the driver's multi-rank launcher (torch.distributed.run, backend nccl = RCCL) with a world
# of one rank on the one-GPU box: env parsing, process-group setup, barriers and the max-over-ranks timing
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02_bench_torchrun_world1.json 2> gpurun_out/r02_bench_torchrun_world1.err; rc=$?
tail -3 gpurun_out/r02_bench_torchrun_world1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02_bench_torchrun_world1.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "n_gpus", "steps", "ms_per_step", "scaling")}, d["config"].get("backend"), d["config"].get("parallelism"))
PY
exit $rc
