#!/usr/bin/env python3
"""gpurun_out/ of scripts/prof_round.sh <tag> -> the committed summaries: profiles/<tag>_kernel_stats.csv, <tag>_pmc_summary.json, <tag>_mfma_util.json,
pmc_traffic.json (scripts/summarize_prof.py), <tag>_{policy,finetune}_kernel_stats.csv and profiles/pmc_traffic_finetune.json (HBM bytes per launch of the
fine-tune step's kernels; units / gfx950 correction as in summarize_prof.py: KiB, FETCH_SIZE doubled)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r4"
subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_prof.py"), tag], check=True, stdout=subprocess.DEVNULL)
if glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_policy_FETCH_SIZE", "*", "*counter_collection.csv")):
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "summarize_policy_pmc.py"), tag], check=False, stdout=subprocess.DEVNULL)
for p in ("policy", "finetune"):
    f = os.path.join(ROOT, "gpurun_out", f"{tag}_{p}_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(ROOT, "profiles", f"{tag}_{p}_kernel_stats.csv"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_finetune_{c}", "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] == c and "arp::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))][c].append(float(r["Counter_Value"]))
kernels = {}
for (k, g), v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        kernels[f"{k} | grid={g}"] = {"kernel": k, "grid_threads": g, "launches": len(v["FETCH_SIZE"]), "fetch_KiB": rd, "write_KiB": wr,
                                     "hbm_bytes_per_launch": (2 * rd + wr) * 1024}
if kernels:
    # the fused weight-gradient + AdamW launches: gemm_nt_kernel<T, float, 0, false, 26, 2>, one grid size per weight shape (M x N / 128^2 workgroups of 256 threads)
    fused = {}
    for e in kernels.values():
        if "gemm_nt_kernel" in e["kernel"] and ", 26, " in e["kernel"]:
            fused[str(e["grid_threads"])] = {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "fetch_KiB": e["fetch_KiB"], "write_KiB": e["write_KiB"],
                                            "weight_elements": e["grid_threads"] // 256 * 128 * 128}
    sha_file = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_csrc_sha1.txt")
    sha = open(sha_file).read().strip() if os.path.exists(sha_file) else None  # the kernel sources these counters were taken on (bench.py: traffic_stale)
    out = {"round": tag, "csrc_sha1": sha, "fused_adamw_gemm_by_grid_threads": fused, "kernels": kernels,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over bench.py --path finetune, scripts/prof_round.sh"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic_finetune.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(fused, indent=1))
