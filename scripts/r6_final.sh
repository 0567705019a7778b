#!/bin/bash
# round 6 closing evidence: full GPU suite, rocprofv3 summaries (kernel stats, PMC passes incl. the policy step's), THEN the driver's command (stdout kept as the driver
# sees it) so that its roofline.traffic comes from counters of this very tree (traffic_stale false), N1 f16c kernel trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
(time timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | grep -E "passed|failed|rror" | head -5) > $O/r6_gpu_suite.txt 2>&1
./scripts/prof_round.sh r6 > $O/prof_round.log 2>&1
python scripts/summarize_round.py r6 > $O/summarize_round.log 2>&1
mkdir -p $O/box_profiles && cp profiles/pmc_traffic.json profiles/pmc_traffic_finetune.json profiles/pmc_traffic_policy.json profiles/r6_pmc_summary.json profiles/r6_mfma_util.json profiles/r6_kernel_stats.csv profiles/r6_policy_kernel_stats.csv profiles/r6_finetune_kernel_stats.csv $O/box_profiles/ 2>/dev/null
(time python bench.py) > $O/r6_bench_default.jsonl 2> $O/r6_bench_default.err
cp $O/bench_full.json $O/r6_bench_full.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r6_n1_trace -- python3 $R/bench.py --path policy --with-encoder --mode f16 --encoder-mode f16c --steps 10 --warmup 3 --cpu-seconds 0 --parity-frames 0 --no-secondary > $R/gpurun_out/prof_r6_n1_trace.log 2>&1
find $R/gpurun_out/prof_r6_n1_trace -name "*kernel_trace.csv" -delete
cp $(find $R/gpurun_out/prof_r6_n1_trace -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r6_n1_f16c_kernel_stats.csv
cd $R
cat $O/r6_gpu_suite.txt; tail -n 1 $O/r6_bench_default.jsonl | head -c 1500; echo; wc -l $O/r6_bench_default.jsonl; tail -3 $O/r6_bench_default.err
