#!/bin/bash
# MFMA-pipe utilisation per kernel of the labelling pass: one --pmc pass (kernel-trace only) over a short bench run; rocprofv3
# serialises the dispatches in counter mode, so every launch is measured alone.  Run on the GPU box via gpurun; summary -> profiles/.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary > $R/gpurun_out/prof_${TAG}_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_sq -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --parity-frames 0 --no-secondary > $R/gpurun_out/prof_${TAG}_sq.log 2>&1
python3 - <<PY
import csv, glob, collections, json, os
R = "$R"; tag = "$TAG"
out = {}
for d in ("mfma", "sq"):
    fs = sorted(glob.glob(f"{R}/gpurun_out/prof_{tag}_{d}/*/*counter_collection.csv"), key=os.path.getmtime)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), v in acc.items():
        if "arp::" not in k:
            continue
        e = out.setdefault(f"{k} | grid={g}", {"kernel": k, "grid_threads": g})
        for c, xs in v.items():
            e[c + "_avg"] = sum(xs) / len(xs)
            e["launches"] = len(xs)
for e in out.values():
    if "SQ_WAVE_CYCLES_avg" in e and e["SQ_WAVE_CYCLES_avg"]:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c + "_avg" in e:
                e[c + "_frac_of_wave_cycles"] = e[c + "_avg"] / e["SQ_WAVE_CYCLES_avg"]
json.dump(out, open(f"{R}/gpurun_out/{tag}_mfma_util.json", "w"), indent=1, sort_keys=True)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("MfmaUtil_avg", 0))[:12]:
    print(round(e.get("MfmaUtil_avg", -1), 1), round(e.get("SQ_WAIT_ANY_frac_of_wave_cycles", -1), 3), e["grid_threads"], k[:110])
PY
