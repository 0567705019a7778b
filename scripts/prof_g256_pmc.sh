#!/bin/bash
# SQ counters of both MFMA shapes of gemm256_nt_kernel from the standalone harness (one --pmc pass, kernel-trace only).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $R/gpurun_out/prof_g256_pmc -- $R/scripts/gemm256_bench.bin > $R/gpurun_out/prof_g256_pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$R/gpurun_out/prof_g256_pmc/*/*counter_collection.csv"))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][-45:], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k] += 1
for k, v in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    if "gemm256" not in k[0]: continue
    n = v.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(x / n, 4) for c, x in v.items() if c != "SQ_WAVE_CYCLES"}, "wave_cycles_per_launch", n / (cnt[k] / 8))
PY
