#!/bin/bash
# Does the A-panel re-fetch of the 256x256 GEMMs (FETCH_SIZE 5.6x the algorithmic read bytes on c_fc) cost time?  The tile walk's group size
# (ARP_GEMM_GROUP_M tile-rows per XCD patch) trades A re-reads against W re-reads: per group size, one --pmc FETCH_SIZE pass (kernel-trace only)
# and one plain timing run of scripts/gemm256_bench.bin.  FETCH_SIZE is in 32-byte units... the summary prints KiB as the counter reports them x 2 (gfx950).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/g256_groupm.txt
rm -f $O
cd /tmp && export TMPDIR=/tmp
for G in 1 2 4 8 16 32; do
  echo "== ARP_GEMM_GROUP_M=$G" >> $O
  ARP_GEMM_GROUP_M=$G $R/scripts/gemm256_bench.bin 2>&1 | grep -E "qkv|c_fc|c_proj|out_proj|4096" | grep -v "32x32" | head -6 >> $O
  export ARP_GEMM_GROUP_M=$G
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_g256_gm$G -- $R/scripts/gemm256_bench.bin > $R/gpurun_out/prof_g256_gm$G.log 2>&1
  unset ARP_GEMM_GROUP_M
  python3 - >> $O <<PY
import csv, glob, collections
fs = sorted(glob.glob("$R/gpurun_out/prof_g256_gm$G/*/*counter_collection.csv"))
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])):
    if "gemm256" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE" and ", false>" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][-60:], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: int(kv[0][1])):
    print("   FETCH (KiB x 2, avg of %d launches) %10.0f MB  grid %s  %s" % (len(v), 2 * sum(v) / len(v) * 1024 / 1e6, k[1], k[0]))
PY
done
cat $O
