#!/bin/bash
# round 5, run 24: row N1 with the N = 257 attention instance free of spills (fragment reads at most six key tiles ahead); old = arp_amd/alt/prev
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run24.txt
rm -f $F
n1() { timeout 600 python bench.py --path policy --with-encoder --cpu-seconds 0 --steps 10 --warmup 3 $@ 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'), d['parity']['max_logit_err_vs_oracle'])" >> $F 2>&1; }
for rep in 1 2; do
echo "-- f16: new / old" >> $F
n1 --mode f16 --encoder-mode f16
ARP_LIB=arp_amd/alt/prev/libarp_hip.so n1 --mode f16 --encoder-mode f16
echo "-- f16c: new / old" >> $F
n1 --mode f16 --encoder-mode f16c
ARP_LIB=arp_amd/alt/prev/libarp_hip.so n1 --mode f16 --encoder-mode f16c
done
echo "== tests" >> $F
(timeout 2400 python -m pytest tests/test_m3ae_gpu.py tests/test_clip_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -5) >> $F
cat $F
