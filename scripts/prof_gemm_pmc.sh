#!/bin/bash
# SQ counters of gemm256_nt_kernel on two shapes (one --pmc pass, kernel-trace only); run on the GPU box via gpurun.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $R/gpurun_out/prof_gemm_pmc -- python3 $R/scripts/gemm_bench.py sq4096 c_fc > $R/gpurun_out/prof_gemm_pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$R/gpurun_out/prof_gemm_pmc/*/*counter_collection.csv"))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:60], r["Grid_Size"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k, v in acc.items():
    if "gemm256" not in k[0]: continue
    n = v.get("SQ_WAVE_CYCLES", 1)
    print(k, {c: round(x / n, 4) for c, x in v.items() if c != "SQ_WAVE_CYCLES"}, "waves_cycles", n)
PY
