#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of scripts/prof_label.sh (gpurun_out/prof_<tag>_*) into the small
files that are committed under profiles/:

  profiles/<tag>_kernel_stats.csv   -- rocprofv3 --kernel-trace --stats summary (as emitted)
  profiles/<tag>_pmc_summary.json   -- FETCH_SIZE / WRITE_SIZE per kernel and grid size (separate
                                       --pmc passes), with the gfx950 corrections of
                                       MI355X_MICROARCH.md (HBM section): counters are KiB;
                                       FETCH_SIZE under-reports a wide coalesced read stream by 2x
  profiles/pmc_traffic.json         -- per bench site: HBM bytes per launch (read by bench.py)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)

# bench site -> substrings that identify its kernel instantiation (round 2: the QKV projection + attention run as qkv_attn_kernel,
# out_proj on the two-workgroups-per-CU kernel; template arguments of gemm256_nt_kernel are <T, OutT, ACT, RESID, SITE, M32> since round 4)
SITES = {
    "vit.c_fc": ("gemm256_nt_kernel", "1, false, 3,"),
    "vit.c_proj": ("gemm256_nt_kernel", "0, true, 4,"),
    "vit.qkv_attn": ("qkv_attn_kernel",),
    "vit.qkv": ("gemm256_nt_kernel", "0, false, 1,"),
    "vit.out_proj": ("gemm2w_kernel", "float, 0, true>"),
    "vit.patch_embed": ("gemm256_nt_kernel", "0, false, 0,"),
    "vit.ln": ("layernorm_kernel",),
}
FRAMES_PER_LAUNCH = int(os.environ.get("FRAMES_PER_LAUNCH", "1024"))  # the largest grid of a site = bench.py's single-stream (isolated) pass: 1024 frames per launch


def one(pattern):
    g = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pattern)), key=os.path.getmtime)
    return g[-1] if g else None  # newest run


stats = one(f"prof_{tag}_trace/*/*_kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))

summary = {}
for counter, d in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    f = one(f"prof_{tag}_{d}/*/*_counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for (k, grid), v in agg.items():
        e = summary.setdefault(f"{k} | grid={grid}", {"kernel": k, "grid_threads": grid})
        e[counter + "_KiB_avg"] = sum(v) / len(v)
        e[counter + "_launches"] = len(v)
for e in summary.values():
    rd = e.get("FETCH_SIZE_KiB_avg")
    wr = e.get("WRITE_SIZE_KiB_avg")
    if rd is not None and wr is not None:
        e["hbm_bytes_per_launch_raw"] = (rd + wr) * 1024
        e["hbm_bytes_per_launch"] = (2 * rd + wr) * 1024  # gfx950: FETCH_SIZE counts 128-B requests as 64 B
json.dump(summary, open(os.path.join(out_dir, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)

traffic = {}
for site, subs in SITES.items():
    cands = [e for e in summary.values() if all(x in e["kernel"] for x in subs) and "hbm_bytes_per_launch" in e]
    if cands:
        e = max(cands, key=lambda x: x["grid_threads"])  # the full-size launches (the class-token-only last block has small grids)
        traffic[site] = {"hbm_bytes_per_launch": e["hbm_bytes_per_launch"], "fetch_KiB": e["FETCH_SIZE_KiB_avg"],
                         "write_KiB": e["WRITE_SIZE_KiB_avg"], "kernel": e["kernel"], "grid_threads": e["grid_threads"],
                         "frames_per_launch": FRAMES_PER_LAUNCH, "round": tag,
                         "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), profiles/{tag}_pmc_summary.json"}
# MFMA-pipe utilisation of the same kernels, measured alone (scripts/prof_mfma.sh -> gpurun_out/<tag>_mfma_util.json): bench.py prints it
# next to the roofline fraction of the dominant kernel
mf = os.path.join(ROOT, "gpurun_out", f"{tag}_mfma_util.json")
if os.path.exists(mf):
    shutil.copy(mf, os.path.join(out_dir, f"{tag}_mfma_util.json"))
    util = json.load(open(mf))
    for site, rec in traffic.items():
        e = util.get(f"{rec['kernel']} | grid={rec['grid_threads']}")
        if e and "MfmaUtil_avg" in e:
            rec["mfma_util_pct"] = round(e["MfmaUtil_avg"], 2)
            rec["sq_wait_any_frac"] = round(e.get("SQ_WAIT_ANY_frac_of_wave_cycles", float("nan")), 3)
sha_file = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_csrc_sha1.txt")
if os.path.exists(sha_file):  # the kernel sources the passes ran on (arp_amd/_srchash.py); bench.py flags a mismatch as traffic_stale
    sha = open(sha_file).read().strip()
    for rec in traffic.values():
        rec["csrc_sha1"] = sha
json.dump(traffic, open(os.path.join(out_dir, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(traffic, indent=1))
