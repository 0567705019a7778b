#!/bin/bash
# One round's rocprofv3 evidence (run on the GPU box via gpurun, then scripts/summarize_round.py here):
#   label path: kernel-trace stats of bench.py --timed-only, FETCH_SIZE / WRITE_SIZE and MfmaUtil / SQ passes (scripts/prof_label.sh, prof_mfma.sh);
#   policy and fine-tune steps: kernel-trace stats; fine-tune step: FETCH_SIZE / WRITE_SIZE passes (counters in their own runs, kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}
python3 $R/arp_amd/_srchash.py > $R/gpurun_out/prof_${TAG}_csrc_sha1.txt  # the sources these counters belong to (bench.py: traffic_stale)
$R/scripts/prof_label.sh $TAG > $R/gpurun_out/prof_${TAG}_label.log 2>&1
$R/scripts/prof_mfma.sh $TAG > $R/gpurun_out/prof_${TAG}_mfma_summary.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for P in policy finetune; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_${P}_trace -- python3 $R/bench.py --path $P --steps 10 --warmup 3 --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}_${P}_trace.log 2>&1
  find $R/gpurun_out/prof_${TAG}_${P}_trace -name "*kernel_trace.csv" -delete
  cp $(find $R/gpurun_out/prof_${TAG}_${P}_trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_${P}_kernel_stats.csv
done
for C in FETCH_SIZE WRITE_SIZE; do  # the policy step's kernels launched eagerly (one dispatch per kernel in the counter file)
  ARP_DT_GRAPH=0 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_policy_$C -- python3 $R/bench.py --path policy --steps 3 --warmup 2 --cpu-seconds 0 --parity-frames 0 > $R/gpurun_out/prof_${TAG}_policy_$C.log 2>&1
  find $R/gpurun_out/prof_${TAG}_policy_$C -name "*kernel_trace.csv" -delete
done
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_finetune_$C -- python3 $R/bench.py --path finetune --steps 3 --warmup 2 --cpu-seconds 0 > $R/gpurun_out/prof_${TAG}_finetune_$C.log 2>&1
  find $R/gpurun_out/prof_${TAG}_finetune_$C -name "*kernel_trace.csv" -delete
done
ls -la $R/gpurun_out/*.csv | tail -5
