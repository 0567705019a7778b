#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
cp arp_amd/libarp_hip.so /tmp/main.so; cp arp_amd/libarp_hip_alt.so /tmp/alt.so
rm -f $O/r4_mix.txt
for v in main alt main alt; do
  cp /tmp/$v.so arp_amd/libarp_hip.so
  echo "== $v (main = v_fma_mix splits, alt = convert / subtract / convert)" >> $O/r4_mix.txt
  python bench.py --path policy --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['sites_ms_per_step']; print('policy', round(d['value']), d['ms_per_step'], s['dt.policy_fwd'], s['dt.image_text_input'])" >> $O/r4_mix.txt
  python bench.py --path policy --with-encoder --mode f32 --encoder-mode f16x3 --no-secondary --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['sites_ms_per_step']; print('x3', round(d['value'],1), d['ms_per_step'], d['parity']['max_logit_err_vs_oracle'], s['m3ae.attn'], s['m3ae.c_fc'], s['m3ae.ln_1'], s['m3ae.ln_2'])" >> $O/r4_mix.txt
done
cp /tmp/main.so arp_amd/libarp_hip.so
cat $O/r4_mix.txt
