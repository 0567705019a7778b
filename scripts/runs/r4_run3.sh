#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 900 python -m pytest tests/test_finetune_gpu.py -q -x -m gpu -k "matches_reference or gradients_match or train_steps or nn_dx" 2>&1 | tail -6) > $O/r4_t_ft2.txt
python bench.py --path finetune --no-secondary --cpu-seconds 0 2>/dev/null > $O/r4_bench_finetune.json
./scripts/prof_round.sh r4 > $O/prof_round.log 2>&1
tail -n 4 $O/r4_t_ft2.txt; python -c "
import json; d=json.load(open('$O/r4_bench_finetune.json')); print(d['value'], d['ms_per_step'], d['roofline']); print(d['sites_ms_per_step'])"
tail -n 5 $O/prof_round.log
