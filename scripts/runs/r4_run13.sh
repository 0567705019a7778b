#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
rm -f $O/r4_iti_wgs.txt
for v in 256 512 768 1024 256 512; do
  echo "== ARP_DT_ITI_WGS=$v" >> $O/r4_iti_wgs.txt
  ARP_DT_ITI_WGS=$v python bench.py --path policy --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['sites_ms_per_step']; print('policy', d['value'], d['ms_per_step'], d['parity']['max_logit_err_vs_oracle'], s['dt.image_text_input'])" >> $O/r4_iti_wgs.txt
done
cat $O/r4_iti_wgs.txt
