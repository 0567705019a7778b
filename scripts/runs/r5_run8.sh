#!/bin/bash
# round 5, run 8: image_text_input's prefetch pinned ahead of the MFMAs (sched_barrier), one or two K-tiles in flight, one or two workgroups per CU
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
F=$O/r5_run8.txt
rm -f $F
pol() { timeout 300 python bench.py --path policy --cpu-seconds 0 --steps 60 --warmup 10 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('top_sites_ms'))" >> $F 2>&1; }
for rep in 1 2; do
echo "-- ahead 2, 256 slices (default)" >> $F; pol
echo "-- ahead 1, 256 slices" >> $F; ARP_DT_ITI_AHEAD=1 pol
echo "-- ahead 1, 512 slices" >> $F; ARP_DT_ITI_AHEAD=1 ARP_DT_ITI_WGS=512 pol
echo "-- ahead 1, 514 slices" >> $F; ARP_DT_ITI_AHEAD=1 ARP_DT_ITI_WGS=514 pol
echo "-- ahead 1, 771 slices" >> $F; ARP_DT_ITI_AHEAD=1 ARP_DT_ITI_WGS=771 pol
done
echo "== tests" >> $F
(timeout 1500 python -m pytest tests/test_policy_gpu.py -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -3) >> $F
cat $F
