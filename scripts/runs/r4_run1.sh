#!/bin/bash
# round 4, GPU batch 1: new tests, LayerNorm-fold A/B in f16, N1 parity probe, online / h5 bench lines
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_clip_gpu.py -q -x -m gpu -k "online_reward_family or online_single" 2>&1 | tail -15) > $O/r4_t_online.txt
(timeout 1200 python -m pytest tests/test_finetune_gpu.py -q -x -m gpu -k "online_adapter or gradientless or leaves_gradient" 2>&1 | tail -15) > $O/r4_t_adapter.txt
for v in default "ARP_QKV_FUSED=0" "ARP_LN_FOLD=1" default "ARP_LN_FOLD=1"; do
  echo "== $v" >> $O/r4_lnfold_ab.txt
  if [ "$v" = default ]; then python bench.py --no-secondary --cpu-seconds 0 --no-alt-bf16 --timed-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'], {k:v for k,v in list(d['sites_ms_per_step'].items())[:9]})" >> $O/r4_lnfold_ab.txt
  else env $v python bench.py --no-secondary --cpu-seconds 0 --no-alt-bf16 --timed-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'], {k:v for k,v in list(d['sites_ms_per_step'].items())[:9]})" >> $O/r4_lnfold_ab.txt; fi
done
(timeout 1500 python scripts/n1_parity_probe.py 8 2>&1 | tail -14) > $O/r4_n1_probe.txt
(python bench.py --path online 2>/dev/null) > $O/r4_bench_online.json
(python bench.py --path h5 2>/dev/null) > $O/r4_bench_h5.json
tail -5 $O/r4_t_online.txt $O/r4_t_adapter.txt; cat $O/r4_lnfold_ab.txt; cat $O/r4_n1_probe.txt; cat $O/r4_bench_online.json | cut -c1-1500; cat $O/r4_bench_h5.json | cut -c1-900
