#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
rm -f $O/r4_cumask.txt
for v in 0 xcd half 0 xcd half; do
  echo "== ARP_CLIP_CUMASK=$v" >> $O/r4_cumask.txt
  ARP_CLIP_CUMASK=$v timeout 300 python bench.py --no-secondary --cpu-seconds 0 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['sites_ms_per_step']; print(round(d['value']), d['ms_per_step'], d['parity']['max_abs_err'] if 'max_abs_err' in d.get('parity',{}) else '', {k:s[k] for k in list(s)[:6]})" >> $O/r4_cumask.txt 2>&1
done
cat $O/r4_cumask.txt
