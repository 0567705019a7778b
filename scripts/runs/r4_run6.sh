#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(timeout 1200 python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "attention" 2>&1 | tail -4) > $O/r4_t_attn.txt
(timeout 1800 python -m pytest tests/test_m3ae_gpu.py tests/test_clip_gpu.py -q -x -m gpu 2>&1 | tail -4) >> $O/r4_t_attn.txt
rm -f $O/r4_n1_modes.txt
for args in "--mode f32 --encoder-mode f16x3" "--mode f32"; do
  echo "== $args" >> $O/r4_n1_modes.txt
  python bench.py --path policy --with-encoder $args --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dtype'], d['parity']['max_logit_err_vs_oracle'], d['roofline']['kernel'], d['roofline']['frac'], d['whole_step'], {k:v for k,v in list(d['sites_ms_per_step'].items())[:12]})" >> $O/r4_n1_modes.txt
done
cat $O/r4_t_attn.txt; cat $O/r4_n1_modes.txt
