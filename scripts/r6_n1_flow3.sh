#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
one() {
  L=$1; shift
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
print('$L', 'ms_per_step', d['ms_per_step'], dict(list(s.items())[:4]))"
}
N1="python bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --steps 16 --warmup 4 --cpu-seconds 0 --no-secondary --parity-frames 0"
{
for rep in 1 2; do
  echo "== rep $rep (two slots, ahead, no gate; churn kinds)"
  one "none            " $N1
  one "full            " $N1 --churn full
  one "malloc          " $N1 --churn malloc
  one "h2d             " $N1 --churn h2d
  one "trainer         " $N1 --churn trainer
  one "encoder         " $N1 --churn encoder
  one "encoder,encfwd  " $N1 --churn encoder,encfwd
  one "trainer,fwd_enc " $N1 --churn trainer,fwd_enc
  one "trainer,encoder " $N1 --churn trainer,encoder
done
} > $O/r6_n1_flow3.txt 2>&1
cut -c1-200 $O/r6_n1_flow3.txt
