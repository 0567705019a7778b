#!/usr/bin/env python3
"""Row N4: single-frame rewards under `rocprofv3 --kernel-trace` -- the kernels of the replayed graph with their start / end stamps.
   python scripts/latency_trace.py [ViT-B/32|ViT-B/16] [calls]            (run it after `--`)
   python scripts/latency_trace.py --analyze <kernel_trace.csv> [calls]   (per-kernel durations and the gaps between kernels)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--analyze":
    import csv, collections
    rows = list(csv.DictReader(open(sys.argv[2])))
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # keep the kernels of the last `calls` passes: a pass starts at a preprocess kernel
    starts = [i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"]]
    starts = starts[-calls:]
    per = collections.defaultdict(lambda: [0, 0.0])
    gaps, spans, busy, counts = [], [], [], []
    for a, b in zip(starts, starts[1:] + [len(rows)]):
        ks = rows[a:b]
        if b == len(rows):  # the last pass: cut at its reward kernel
            for j, r in enumerate(ks):
                if "reward_kernel" in r["Kernel_Name"]:
                    ks = ks[:j + 1]
                    break
        t0, t1 = int(ks[0]["Start_Timestamp"]), int(ks[-1]["End_Timestamp"])
        spans.append((t1 - t0) / 1e3)
        busy.append(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e3)
        counts.append(len(ks))
        for r, nx in zip(ks, ks[1:]):
            gaps.append((int(nx["Start_Timestamp"]) - int(r["End_Timestamp"])) / 1e3)
        for r in ks:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("arp::", "").split("(")[0][:70]
            if "skinny_gemm" in name:
                name += f" grid {r.get('Grid_Size_X', '?')}x{r.get('Grid_Size_Y', '?')} wg {r.get('Workgroup_Size_X', '?')}"
            per[name][0] += 1
            per[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = len(spans)
    print(f"{n} passes: {np.mean(counts):.0f} kernels, first start -> last end {np.mean(spans):.1f} us (min {np.min(spans):.1f}); "
          f"kernels busy {np.mean(busy):.1f} us; gap between kernels mean {np.mean(gaps):.2f} us, median {np.median(gaps):.2f} us")
    for name, (cnt, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print(f"  {us / n:8.1f} us/pass  {cnt / n:5.1f} launches  {us / cnt:6.2f} us each  {name}")
    sys.exit(0)

from arp_amd import clip, synth
name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = clip.MODELS[name]
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=64, n_streams=1).set_text(synth.prompt_tokens(1, 8, seed=2))
fr = synth.procgen_like_frames(4, seed=3)
for _ in range(5):
    m.label(fr[:1])
t0 = time.perf_counter()
for _ in range(calls):
    m.label(fr[:1])
print(f"{name}: {(time.perf_counter() - t0) / calls * 1e3:.3f} ms per single-frame call (under the tracer)")
m.close()
