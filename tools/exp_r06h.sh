#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export HSA_ENABLE_IPC_MODE_LEGACY=0
SECONDS=0; timeout 600 python bench.py --gpus 8 --share-gpu --steps 18 --warmup 2 --no-f16 --no-cpu-baseline > gpurun_out/r06h_gpus8.json 2> gpurun_out/r06h_gpus8.err
echo rc=$? elapsed=$SECONDS; tail -5 gpurun_out/r06h_gpus8.err; tail -c 600 gpurun_out/r06h_gpus8.json
python - <<PY
import json
j=json.loads(open("gpurun_out/r06h_gpus8.json").read().strip().splitlines()[-1]); print(j["n_gpus"], j["value"], j.get("train_dp"))
PY
