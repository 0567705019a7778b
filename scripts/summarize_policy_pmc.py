#!/usr/bin/env python3
"""gpurun_out/prof_r2_policy_{FETCH,WRITE}_SIZE (scripts/prof_policy_pmc.sh) -> profiles/pmc_traffic_policy.json: HBM bytes per launch of
every kernel of the policy train step, and per call site for the sites bench.py --path policy names in its roofline block.  Counter
units and the gfx950 correction as in scripts/summarize_prof.py (KiB; FETCH_SIZE under-reports a wide coalesced read stream by 2x)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_policy_{c}", "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        raise SystemExit(f"no counter_collection.csv for {c}")
    for r in csv.DictReader(open(fs[-1])):
        if r["Counter_Name"] == c and "arp::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))][c].append(float(r["Counter_Value"]))
kernels = {}
for (k, g), v in acc.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        rd, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        kernels[f"{k} | grid={g}"] = {"kernel": k, "grid_threads": g, "launches": len(v["FETCH_SIZE"]), "fetch_KiB": rd, "write_KiB": wr,
                                     "hbm_bytes_per_launch": (2 * rd + wr) * 1024}
SITES = {  # call site of bench.py --path policy -> kernels launched once per step there
    "dt.clip_adam": ("norms_partial_kernel", "adam_kernel"),
    "dt.adapter_dy_fused": ("adapter_dy_kernel",),
    "dt.adapter_fc_dW": ("gemm_tn256_kernel",), "dt.adapter_fc2_dW": ("gemm_tn256_kernel",), "dt.adapter_fc1_dW": ("gemm_tn256_kernel",),
    "dt.image_text_input": ("iti_x3_kernel",), "dt.policy_fwd": ("policy_fused_kernel",),
}
sites = {}
for site, subs in SITES.items():
    tot, parts = 0.0, {}
    for sub in subs:
        cands = [e for e in kernels.values() if sub in e["kernel"]]
        if cands:
            e = max(cands, key=lambda x: x["hbm_bytes_per_launch"])
            parts[sub] = e["hbm_bytes_per_launch"]
            tot += e["hbm_bytes_per_launch"]
    if parts:
        sites[site] = {"hbm_bytes_per_launch": tot, "kernels": parts,
                       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager launches), scripts/prof_policy_pmc.sh"}
sha_file = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_csrc_sha1.txt")
out = {"round": tag, "csrc_sha1": open(sha_file).read().strip() if os.path.exists(sha_file) else None, "sites": sites, "kernels": kernels}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic_policy.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(sites, indent=1))
