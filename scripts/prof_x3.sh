#!/bin/bash
# MFMA-pipe utilisation, wait fractions and LDS bank conflicts of the (hi, lo) binary16 kernels of round 4 -- the f16x3 encoder step (attn_x3_kernel, the
# K-concatenated GEMMs) and the policy step (policy_fused_kernel<..., true>, iti_x3_kernel): --pmc passes with kernel-trace only, one counter group per run.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}
cd /tmp && export TMPDIR=/tmp
ARGS_ENC="--path policy --with-encoder --mode f32 --encoder-mode f16x3 --steps 2 --warmup 1 --cpu-seconds 0 --no-secondary"
ARGS_POL="--path policy --steps 3 --warmup 2 --cpu-seconds 0 --no-secondary"
for W in enc pol; do
  if [ $W = enc ]; then A="$ARGS_ENC"; else A="$ARGS_POL"; fi
  rocprofv3 --pmc MfmaUtil --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_x3_${W}_mfma -- python3 $R/bench.py $A > $R/gpurun_out/prof_${TAG}_x3_${W}_mfma.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_x3_${W}_sq -- python3 $R/bench.py $A > $R/gpurun_out/prof_${TAG}_x3_${W}_sq.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_x3_${W}_lds -- python3 $R/bench.py $A > $R/gpurun_out/prof_${TAG}_x3_${W}_lds.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json, os
R = "$R"; tag = "$TAG"
out = {}
for w in ("enc", "pol"):
    for d in ("mfma", "sq", "lds"):
        fs = sorted(glob.glob(f"{R}/gpurun_out/prof_{tag}_x3_{w}_{d}/*/*counter_collection.csv"), key=os.path.getmtime)
        if not fs:
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fs[-1])):
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for (k, g), v in acc.items():
            if "arp::" not in k:
                continue
            e = out.setdefault(f"{w} | {k} | grid={g}", {"workload": w, "kernel": k, "grid_threads": g})
            for c, xs in v.items():
                e[c + "_avg"] = sum(xs) / len(xs)
                e["launches"] = len(xs)
for e in out.values():
    if e.get("SQ_WAVE_CYCLES_avg"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if c + "_avg" in e:
                e[c + "_frac_of_wave_cycles"] = e[c + "_avg"] / e["SQ_WAVE_CYCLES_avg"]
    if e.get("SQ_LDS_IDX_ACTIVE_avg"):
        e["lds_bank_conflict_frac_of_lds_active"] = e.get("SQ_LDS_BANK_CONFLICT_avg", 0.0) / e["SQ_LDS_IDX_ACTIVE_avg"]
json.dump(out, open(f"{R}/gpurun_out/{tag}_x3_pmc.json", "w"), indent=1, sort_keys=True)
for k, e in sorted(out.items(), key=lambda kv: -kv[1].get("MfmaUtil_avg", 0))[:24]:
    print(round(e.get("MfmaUtil_avg", -1), 1), round(e.get("SQ_WAIT_ANY_frac_of_wave_cycles", -1), 3), round(e.get("lds_bank_conflict_frac_of_lds_active", -1), 3), e["grid_threads"], k[:130])
PY
