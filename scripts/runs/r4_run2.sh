#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_finetune_gpu.py -q -x -m gpu 2>&1 | tail -15) > $O/r4_t_ft.txt
for v in "ARP_FT_NN=1" "ARP_FT_NN=0" "ARP_FT_NN=1" "ARP_FT_NN=0"; do
  echo "== $v" >> $O/r4_ft_nn_ab.txt
  env $v python bench.py --path finetune --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['final_aux'], {k:v for k,v in list(d['sites_ms_per_step'].items())[:14]})" >> $O/r4_ft_nn_ab.txt
done
(python bench.py --path online 2>/dev/null) > $O/r4_bench_online.json
(time timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -12) > $O/r4_t_all.txt 2>&1
tail -n 8 $O/r4_t_ft.txt; cat $O/r4_ft_nn_ab.txt; python -c "
import json; d=json.load(open('$O/r4_bench_online.json')); print(d['reward'], d['greedy_action'], d['more_rewards_ms'])"; tail -n 14 $O/r4_t_all.txt
