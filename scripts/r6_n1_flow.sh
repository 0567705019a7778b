#!/bin/bash
# Why is `bench.py --path policy --with-encoder` 1.8 ms per step faster with its parity gate than without (r6 third run, same box)?  One box, interleaved.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
one() {
  L=$1; shift
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
s=d.get('top_sites_ms') or {}
print('$L', 'ms_per_step', d['ms_per_step'], dict(list(s.items())[:7]))"
}
N1="python bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --steps 16 --warmup 4 --cpu-seconds 0 --no-secondary"
{
for rep in 1 2; do
  echo "== rep $rep"
  one "gate (parity-frames 8), two slots, ahead      " $N1
  one "no gate,               two slots, ahead      " $N1 --parity-frames 0
  one "no gate + churn,       two slots, ahead      " $N1 --parity-frames 0 --churn
  one "no gate, HW queues 16, two slots, ahead      " env GPU_MAX_HW_QUEUES=16 $N1 --parity-frames 0
  one "no gate, HW queues 4,  two slots, ahead      " env GPU_MAX_HW_QUEUES=4 $N1 --parity-frames 0
  one "no gate,               single slot (r5 flow) " $N1 --parity-frames 0 --single-slot
  one "gate,                  single slot (r5 flow) " $N1 --single-slot
  one "no gate, single slot, captured encoder       " env ARP_DT_ENC_EAGER=0 $N1 --parity-frames 0 --single-slot
  one "no gate, two slots, at head                  " $N1 --parity-frames 0 --no-encode-ahead
  one "gate,    two slots, at head                  " $N1 --no-encode-ahead
done
} > $O/r6_n1_flow.txt 2>&1
cut -c1-260 $O/r6_n1_flow.txt
