#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
PER_SEED=1 python scripts/adapter_plan_gpu.py off 22e 22h > gpurun_out/r6_floor_per_seed.txt 2>&1
cat gpurun_out/r6_floor_per_seed.txt | cut -c1-400
