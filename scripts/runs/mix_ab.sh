#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 1800 python -m pytest tests/test_policy_gpu.py tests/test_m3ae_gpu.py tests/test_ops_gpu.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|assert" | tail -6) > $O/r4_t_mix.txt
rm -f $O/r4_mix.txt
python bench.py --path policy --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['sites_ms_per_step']; print('policy', d['value'], d['ms_per_step'], d['parity']['max_logit_err_vs_oracle'], s['dt.policy_fwd'], s['dt.image_text_input'])" >> $O/r4_mix.txt
python bench.py --path policy --with-encoder --mode f32 --encoder-mode f16x3 --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity']['max_logit_err_vs_oracle'], {k:v for k,v in list(d['sites_ms_per_step'].items())[:8]})" >> $O/r4_mix.txt
cat $O/r4_t_mix.txt; cat $O/r4_mix.txt
