#!/bin/bash
# stream priorities in the encoder-inside step: encoder streams high / policy compute stream low
cd ${GRAFT_REPO_ROOT:-/root/repo}
N1="python bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --steps 16 --warmup 4 --cpu-seconds 0 --no-secondary"
one() { L=$1; shift; "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$L', d['ms_per_step'], (d.get('parity') or {}).get('max_logit_err_vs_oracle'))"; }
for rep in 1 2 3; do
  one "default (all normal)        " $N1
  one "encoder high                " env ARP_ENC_PRIO=-1 $N1
  one "policy low                  " env ARP_DT_PRIO=1 $N1
  one "encoder high + policy low   " env ARP_ENC_PRIO=-1 ARP_DT_PRIO=1 $N1
  one "encoder low (control)       " env ARP_ENC_PRIO=1 $N1
  one "policy high (control)       " env ARP_DT_PRIO=-1 $N1
done
